"""Path post-processing and profile counting around the pairwise aligner (host side, integer logic).

Vectorised restatements of praline/util/align.py:187-305 and praline/util/support.py:32-40; the
numeric fill / traceback itself lives in the HIP library (native.py)."""
import numpy as np

# traceback flag bits (praline/util/align.py:15-21, praline/util/cext.c:9-15)
TRACEBACK_MATCH_MATCH = 1 << 1
TRACEBACK_MATCH_INSERT_UP = 1 << 2
TRACEBACK_MATCH_INSERT_LEFT = 1 << 3
TRACEBACK_INSERT_UP_OPEN = 1 << 4
TRACEBACK_INSERT_UP_EXTEND = 1 << 5
TRACEBACK_INSERT_LEFT_OPEN = 1 << 6
TRACEBACK_INSERT_LEFT_EXTEND = 1 << 7


def window(l, size=2):
    """Index tuples of a sliding window (support.py:32-40)."""
    for n in range(len(l) - size + 1):
        yield tuple(n + m for m in range(size))


def _advances(path):
    path = np.asarray(path)
    return (path[1:] - path[:-1]) > 0


def get_frequencies(alignment, trid):
    """Per alignment column, the symbol counts of the sequences that advance in that column
    (util/align.py:187-213); rows of -1 (local padding) never advance.  One scatter-add over all (column, sequence)
    advances (the reference loops over columns and sequences in Python)."""
    path = np.asarray(alignment.path)
    tracks = [seq.get_track(trid) for seq in alignment.items]
    freqs = np.zeros((path.shape[0] - 1, tracks[0].alphabet.size), dtype=int)
    rows, cols = np.nonzero(_advances(path))
    if len(rows):
        offsets = np.concatenate([[0], np.cumsum([len(t.values) for t in tracks])[:-1]])
        values = np.concatenate([np.asarray(t.values) for t in tracks])
        np.add.at(freqs, (rows, values[offsets[cols] + path[rows + 1, cols] - 1]), 1)
    return freqs


def compress_path(path, compress_idx):
    """Drop the rows in which the master sequence does not advance (util/align.py:215-232)."""
    path = np.asarray(path)
    keep = np.concatenate([[0], np.nonzero(_advances(path)[:, compress_idx])[0] + 1])
    return path[keep, :]


def extend_path_local(path, extend_length, extend_idx):
    """Pad a local path to the full extent of sequence extend_idx with -1 rows
    (util/align.py:234-266)."""
    path = np.asarray(path)
    first_idx = path[0, extend_idx]
    last_idx = path[-1, extend_idx]
    parts = []
    if first_idx > 0:
        ext = np.full((first_idx, path.shape[1]), -1, dtype=int)
        ext[:, extend_idx] = np.arange(first_idx)
        parts.append(ext)
    parts.append(path)
    if last_idx < extend_length:
        ext = np.full((extend_length - last_idx, path.shape[1]), -1, dtype=int)
        ext[:, extend_idx] = np.arange(last_idx + 1, extend_length + 1)
        parts.append(ext)
    return np.vstack(parts)


def extend_path_semiglobal(path, mat_shape):
    """Extend a semiglobal path to the matrix corners (util/align.py:268-297).  The batched device
    path already returns extended paths; this host twin serves RawPairwiseAligner callers."""
    path = np.asarray(path)
    n, m = mat_shape
    parts = []
    if path[0, 0] != 0:
        pre = np.zeros((path[0, 0], 2), dtype=int)
        pre[:, 0] = np.arange(path[0, 0])
        parts.append(pre)
    elif path[0, 1] != 0:
        pre = np.zeros((path[0, 1], 2), dtype=int)
        pre[:, 1] = np.arange(path[0, 1])
        parts.append(pre)
    parts.append(path)
    if path[-1, 0] != n - 1:
        post = np.empty(((n - 1) - path[-1, 0], 2), dtype=int)
        post[:, 1] = path[-1, 1]
        post[:, 0] = np.arange(path[-1, 0] + 1, n)
        parts.append(post)
    elif path[-1, 1] != m - 1:
        post = np.empty(((m - 1) - path[-1, 1], 2), dtype=int)
        post[:, 0] = path[-1, 0]
        post[:, 1] = np.arange(path[-1, 1] + 1, m)
        parts.append(post)
    return np.vstack(parts)


def auto_align_mode(one, two):
    """util/align.py:299-305"""
    return "semiglobal_one" if len(one) > len(two) else "semiglobal_two"


def zero_idxs_to_rectangles(zero_idxs, max_rects=1024):
    """The Waterman-Eggert masks the reference builds are full rectangles appended one after the
    other, enumerated first-coordinate-major (praline/component/preprofile.py:247-255).  Parse such a
    list back into inclusive rectangles (y0, y1, x0, x1) - any cell list decomposes into row-run
    rectangles, the batched plans take any number of them (up to 4 in registers, more through per-row
    column masks); returns None only beyond max_rects (the caller then falls back to the dense-mask
    raw path)."""
    idx = np.asarray(list(zero_idxs), dtype=np.int64).reshape(-1, 2)
    rects = []
    i, n = 0, idx.shape[0]
    while i < n:
        y0, x0 = idx[i]
        # extent along the second coordinate: run with constant first coordinate
        j = i
        while j + 1 < n and idx[j + 1, 0] == y0 and idx[j + 1, 1] == idx[j, 1] + 1:
            j += 1
        w = j - i + 1
        x1 = x0 + w - 1
        # how many full rows of that shape follow
        rows = 1
        while True:
            s = i + rows * w
            if s + w > n:
                break
            blk = idx[s:s + w]
            if not (np.all(blk[:, 0] == y0 + rows) and np.array_equal(blk[:, 1], np.arange(x0, x1 + 1))):
                break
            rows += 1
        rects.append((int(y0), int(y0 + rows - 1), int(x0), int(x1)))
        i += rows * w
        if len(rects) > max_rects:
            return None
    return rects
