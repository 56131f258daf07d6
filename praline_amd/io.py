"""Score-matrix loader, FASTA in, aligned FASTA out - the I/O functions the MSA pipeline needs at its ends
(praline/__init__.py:57-102 open_builtin / load_score_matrix, 105-136 load_sequence_fasta, 252-304
write_alignment_fasta).  Host-side text handling; nothing here touches the device."""
import io as _io
import uuid as _uuid

import numpy as np

from .container import Alphabet, PlainTrack, ScoreMatrix, Sequence, TRACK_ID_INPUT
from .core import DataError


def open_builtin(name):
    """A packaged resource as a text file object - 'matrices/<table>' for the score tables the reference ships
    (praline/__init__.py:57-65, praline/matrices/*): blosum30 ... blosum100, nucleotide."""
    from .matrices import builtin_text
    prefix = "matrices/"
    if not name.startswith(prefix):
        raise DataError("unknown builtin resource '{0}'".format(name))
    try:
        return _io.StringIO(builtin_text(name[len(prefix):]))
    except KeyError as e:
        raise DataError(str(e))


def load_score_matrix(f, alphabet=None, encoding="utf-8"):
    """ScoreMatrix from the reference's text format (praline/__init__.py:67-102): '#' starts a comment, the first
    non-empty line lists the column symbols, every further line is a row symbol followed by its scores (values beyond the
    listed columns are ignored).  Without `alphabet` one is made from the column symbols in file order, as the
    reference does (its id carries a fresh uuid).  `f`: path or file object (text or bytes)."""
    handle = open(f, "rb") if isinstance(f, str) else f
    try:
        raw = handle.read()
    finally:
        if isinstance(f, str):
            handle.close()
    text = raw.decode(encoding) if isinstance(raw, bytes) else raw
    rows = []
    for line in text.splitlines():
        cut = line.find("#")
        if cut >= 0:
            line = line[:cut]
        line = line.strip()
        if line:
            rows.append(line.split())
    if not rows:
        raise DataError("empty score matrix")
    columns = rows[0]
    scores = {}
    for row in rows[1:]:
        for i, value in enumerate(row[1:]):
            if i < len(columns):
                scores[row[0], columns[i]] = float(value)
    if not alphabet:
        alphabet = Alphabet("__anonymous_from_matrix_{0}__".format(_uuid.uuid4().hex), [(s, i) for i, s in enumerate(columns)])
    return ScoreMatrix(scores, [alphabet, alphabet])


def load_sequence_fasta(source, alphabet, track_id=TRACK_ID_INPUT):
    """Sequences of a FASTA file (path or text file object), residues upper-cased and mapped through
    `alphabet`; the record name is the header up to the first whitespace."""
    handle = open(source, "r") if isinstance(source, str) else source
    try:
        records = []
        name, chunks = None, []
        for line in handle:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if name is not None:
                    records.append((name, "".join(chunks)))
                fields = line[1:].split()
                name, chunks = (fields[0] if fields else ""), []
            elif name is not None:
                chunks.append(line)
        if name is not None:
            records.append((name, "".join(chunks)))
    finally:
        if isinstance(source, str):
            handle.close()
    out = []
    for name, residues in records:
        track = PlainTrack([c for c in residues.upper()], alphabet)
        out.append(Sequence(name, [(track_id, track)]))
    return out


def alignment_rows(alignment, track_id=TRACK_ID_INPUT):
    """One gapped string per aligned sequence: column i shows the residue a sequence consumes between
    path rows i and i+1, '-' where it does not advance."""
    path = np.asarray(alignment.path)
    rows = []
    for j, sequence in enumerate(alignment.items):
        track = sequence.get_track(track_id)
        if track.tid != PlainTrack.tid:
            raise DataError("can only write FASTA alignments for plain tracks")
        if np.any(path[:, j] == -1):
            raise DataError("the FASTA format does not currently support local alignments")
        advance = (path[1:, j] - path[:-1, j]) > 0
        symbols = np.array([track.alphabet.index_to_symbol(int(v)) for v in track.values] + ["-"])
        idx = np.where(advance, path[1:, j] - 1, len(track.values))
        rows.append("".join(symbols[idx]))
    return rows


def write_alignment_fasta(target, alignment, track_id=TRACK_ID_INPUT, line_length=72):
    """Aligned FASTA: '>name' (cut to line_length - 1 characters), then the gapped sequence wrapped
    at line_length columns, every line newline-terminated."""
    text = _io.StringIO()
    for sequence, row in zip(alignment.items, alignment_rows(alignment, track_id)):
        text.write(">{0}\n".format(sequence.name[:line_length - 1]))
        for k in range(0, len(row), line_length):
            text.write(row[k:k + line_length] + "\n")
    data = text.getvalue()
    if isinstance(target, str):
        with open(target, "w", encoding="utf-8") as f:
            f.write(data)
    else:
        target.write(data)
    return data
