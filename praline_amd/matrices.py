"""Score matrices shipped with the package.

BLOSUM62 (Henikoff & Henikoff 1992; public NCBI table, 1/2-bit units) stored as its lower
triangle in the NCBI symbol order and expanded into the 27-symbol amino-acid alphabet of the
operator layer (container.ALPHABET_AA, same symbol order as praline/container/alphabet.py:100-104
so PlainTrack index arrays are interchangeable).  Rows/columns of symbols the table does not
define (U, O, J) stay zero, exactly as praline.load_score_matrix leaves them
(praline/__init__.py:67-102).
"""
import numpy as np

AA_SYMBOLS = "ARNDCEQGHILKMFPSTWYVUOBZJX*"
_BLOSUM62_ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"
_BLOSUM62_LOWER = """
4
-1 5
-2 0 6
-2 -2 1 6
0 -3 -3 -3 9
-1 1 0 0 -3 5
-1 0 0 2 -4 2 5
0 -2 0 -1 -3 -2 -2 6
-2 0 1 -1 -3 0 0 -2 8
-1 -3 -3 -3 -1 -3 -3 -4 -3 4
-1 -2 -3 -4 -1 -2 -3 -4 -3 2 4
-1 2 0 -1 -3 1 1 -2 -1 -3 -2 5
-1 -1 -2 -3 -1 0 -2 -3 -2 1 2 -1 5
-2 -3 -3 -3 -2 -3 -3 -3 -1 0 0 -3 0 6
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4 7
1 -1 1 0 -1 0 0 0 -1 -2 -2 0 -1 -2 -1 4
0 -1 0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1 1 5
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1 1 -4 -3 -2 11
-2 -2 -2 -3 -2 -1 -2 -3 2 -1 -1 -2 -1 3 -3 -2 -2 2 7
0 -3 -3 -3 -1 -2 -2 -3 -3 3 1 -2 1 -1 -2 -2 0 -3 -1 4
-2 -1 3 4 -3 0 1 -1 0 -3 -4 0 -3 -3 -2 0 -1 -4 -3 -3 4
-1 0 0 1 -3 3 4 -2 0 -3 -3 1 -1 -3 -1 0 -1 -3 -2 -2 1 4
0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2 0 0 -2 -1 -1 -1 -1 -1
-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 1
"""


def blosum62_matrix():
    """float32 [27, 27] BLOSUM62 in ALPHABET_AA index order."""
    m = np.zeros((len(AA_SYMBOLS), len(AA_SYMBOLS)), dtype=np.float32)
    rows = [r.split() for r in _BLOSUM62_LOWER.strip().splitlines()]
    for i, row in enumerate(rows):
        a = AA_SYMBOLS.index(_BLOSUM62_ORDER[i])
        for j, v in enumerate(row):
            b = AA_SYMBOLS.index(_BLOSUM62_ORDER[j])
            m[a, b] = m[b, a] = float(v)
    return m


DNA_SYMBOLS = "ATGCSWRYKMBVHDN"
# NUC.4.4 (public NCBI nucleotide table incl. IUPAC ambiguity codes), lower triangle in
# ALPHABET_DNA order (praline/container/alphabet.py:106-109).
_NUC44_LOWER = """
5
-4 5
-4 -4 5
-4 -4 -4 5
-4 -4 1 1 -1
1 1 -4 -4 -4 -1
1 -4 1 -4 -2 -2 -1
-4 1 -4 1 -2 -2 -4 -1
-4 1 1 -4 -2 -2 -2 -2 -1
1 -4 -4 1 -2 -2 -2 -2 -4 -1
-4 -1 -1 -1 -1 -3 -3 -1 -1 -3 -1
-1 -4 -1 -1 -1 -3 -1 -3 -3 -1 -2 -1
-1 -1 -4 -1 -3 -1 -3 -1 -3 -1 -2 -2 -1
-1 -1 -1 -4 -3 -1 -1 -3 -1 -3 -2 -2 -2 -1
-2 -2 -2 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1 -1
"""


def nucleotide_matrix():
    """float32 [15, 15] NUC.4.4 in ALPHABET_DNA index order (A T G C are indices 0-3)."""
    m = np.zeros((15, 15), dtype=np.float32)
    for i, row in enumerate(r.split() for r in _NUC44_LOWER.strip().splitlines()):
        for j, v in enumerate(row):
            m[i, j] = m[j, i] = float(v)
    return m


def builtin_names():
    """Names of the packaged score tables (the reference's praline/matrices directory: blosum30 ... blosum100, nucleotide)."""
    from .matrix_tables import TABLES
    return sorted(TABLES)


def builtin_text(name):
    """A packaged table in the reference's text format (header row of column symbols, then one row per symbol:
    the symbol followed by its scores), as praline.load_score_matrix reads it (praline/__init__.py:67-102)."""
    from .matrix_tables import TABLES
    if name not in TABLES:
        raise KeyError("no packaged score matrix named %r (have: %s)" % (name, ", ".join(sorted(TABLES))))
    symbols, tri = TABLES[name]
    rows = [r.split() for r in tri.strip().splitlines()]
    n = len(symbols)
    full = [[rows[max(i, j)][min(i, j)] for j in range(n)] for i in range(n)]
    width = max(len(v) for r in full for v in r) + 1
    lines = ["# %s (values as packaged with PRALINE: public NCBI table)" % name,
             " " + "".join(s.rjust(width) for s in symbols)]
    for i, s in enumerate(symbols):
        lines.append(s + "".join(v.rjust(width) for v in full[i]))
    return "\n".join(lines) + "\n"
