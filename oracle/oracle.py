"""TEST INFRASTRUCTURE ONLY - ctypes front end of the CPU oracle (oracle/praline_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (praline_amd) must never do so.  The function names and argument shapes mirror
the reference's native functions (praline/util/cext.c:506-520) and the python glue around them
(praline/component/align.py:302-447) so parity tests read like reference usage.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpraline_oracle.so")

MODES = {"global": 0, "local": 1, "semiglobal_both": 2, "semiglobal_one": 3,
         "semiglobal_two": 4}

# praline/util/align.py:15-21
TRACEBACK_MATCH_MATCH = 1 << 1
TRACEBACK_MATCH_INSERT_UP = 1 << 2
TRACEBACK_MATCH_INSERT_LEFT = 1 << 3
TRACEBACK_INSERT_UP_OPEN = 1 << 4
TRACEBACK_INSERT_UP_EXTEND = 1 << 5
TRACEBACK_INSERT_LEFT_OPEN = 1 << 6
TRACEBACK_INSERT_LEFT_EXTEND = 1 << 7


def build(force=False):
    """Compile the oracle (gcc, seconds).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "praline_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libpraline_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.oracle_build_nonzero.argtypes = [vp, i64, i64, vp]
        L.oracle_build_nonzero.restype = None
        L.oracle_build_scores.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]
        L.oracle_build_scores.restype = None
        L.oracle_build_scores_fma.argtypes = [i32, vp, vp, vp, vp, vp, i64, i64, vp]
        L.oracle_build_scores_fma.restype = None
        L.oracle_init_boundaries.argtypes = [i32, vp, vp, vp, vp, i64, i64]
        L.oracle_init_boundaries.restype = None
        L.oracle_align_fill.argtypes = [i32, vp, vp, vp, vp, vp, vp, i64, i64]
        L.oracle_align_fill.restype = None
        L.oracle_end_cell.argtypes = [i32, vp, i64, i64, vp, vp]
        L.oracle_end_cell.restype = None
        L.oracle_traceback.argtypes = [vp, i64, i64, vp, vp]
        L.oracle_traceback.restype = i64
        L.oracle_extend_path_semiglobal.argtypes = [vp, i64, i64, i64, vp]
        L.oracle_extend_path_semiglobal.restype = i64
        L.oracle_pairwise.argtypes = [i32, vp, vp, vp, i64, i64, i64, f32, f32, vp, i64, vp, vp]
        L.oracle_pairwise.restype = i64
        L.oracle_batch_scores.argtypes = [i32, vp, vp, vp, vp, i64, vp, i64, f32, f32, i32, vp]
        L.oracle_batch_scores.restype = i32
        L.oracle_batch_align.argtypes = [i32, vp, vp, vp, vp, vp, i64, vp, i64, f32, f32, vp, vp, i32, vp, vp, vp, vp]
        L.oracle_batch_align.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr_array(arrs):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def build_nonzero_matrix(i):
    """praline/component/align.py:449-458"""
    i = _c(i, np.float32)
    nz = np.empty(i.shape, dtype=np.int64)
    lib().oracle_build_nonzero(_p(i), i.shape[0], i.shape[1], _p(nz))
    return nz


def cext_build_scores(i1s, i2s, i1nzs, i2nzs, ss, m):
    """praline/util/cext.c:308-455 - fills m in place (must be C-contiguous float32)."""
    assert m.flags.c_contiguous and m.dtype == np.float32
    i1s = [_c(a, np.float32) for a in i1s]
    i2s = [_c(a, np.float32) for a in i2s]
    i1nzs = [_c(a, np.int64) for a in i1nzs]
    i2nzs = [_c(a, np.int64) for a in i2nzs]
    ss = [_c(a, np.float32) for a in ss]
    A1 = np.array([a.shape[1] for a in i1s], dtype=np.int64)
    A2 = np.array([a.shape[1] for a in i2s], dtype=np.int64)
    lib().oracle_build_scores(len(i1s), _ptr_array(i1s), _ptr_array(i2s), _ptr_array(i1nzs),
                              _ptr_array(i2nzs), _ptr_array(ss), _p(A1), _p(A2),
                              i1s[0].shape[0], i2s[0].shape[0], _p(m))


def build_scores_fma(i1s, i2s, ss):
    """m = sum_sets P1.S.P2^T in the HIP kernels' evaluation order (fp32 fma chains)."""
    i1s = [_c(a, np.float32) for a in i1s]
    i2s = [_c(a, np.float32) for a in i2s]
    ss = [_c(a, np.float32) for a in ss]
    A1 = np.array([a.shape[1] for a in i1s], dtype=np.int64)
    A2 = np.array([a.shape[1] for a in i2s], dtype=np.int64)
    m = np.zeros((i1s[0].shape[0], i2s[0].shape[0]), dtype=np.float32)
    lib().oracle_build_scores_fma(len(i1s), _ptr_array(i1s), _ptr_array(i2s), _ptr_array(ss),
                                  _p(A1), _p(A2), m.shape[0], m.shape[1], _p(m))
    return m


def _cext_align(mode, m, g1, g2, o, t, z):
    for a, dt in ((m, np.float32), (g1, np.float32), (g2, np.float32), (o, np.float32),
                  (t, np.uint8), (z, np.uint8)):
        assert a.flags.c_contiguous and a.dtype == dt
    lib().oracle_align_fill(MODES[mode], _p(m), _p(g1), _p(g2), _p(o), _p(t), _p(z),
                            m.shape[0], m.shape[1])


def cext_align_global(m, g1, g2, o, t, z):
    _cext_align("global", m, g1, g2, o, t, z)


def cext_align_local(m, g1, g2, o, t, z):
    _cext_align("local", m, g1, g2, o, t, z)


def cext_align_semiglobal_both(m, g1, g2, o, t, z):
    _cext_align("semiglobal_both", m, g1, g2, o, t, z)


def cext_align_semiglobal_one(m, g1, g2, o, t, z):
    _cext_align("semiglobal_one", m, g1, g2, o, t, z)


def cext_align_semiglobal_two(m, g1, g2, o, t, z):
    _cext_align("semiglobal_two", m, g1, g2, o, t, z)


CEXT_ALIGN_FUNCTIONS = {"local": cext_align_local, "global": cext_align_global,
                        "semiglobal_both": cext_align_semiglobal_both,
                        "semiglobal_one": cext_align_semiglobal_one,
                        "semiglobal_two": cext_align_semiglobal_two}


def init_matrices(mode, g1, g2, zero_idxs=None):
    """praline/component/align.py:357-385 - returns freshly initialised (o, t, z)."""
    g1 = _c(g1, np.float32)
    g2 = _c(g2, np.float32)
    L1, L2 = g1.shape[0], g2.shape[0]
    o = np.zeros((L1 + 1, L2 + 1, 3), dtype=np.float32)
    t = np.zeros((L1 + 1, L2 + 1, 3), dtype=np.uint8)
    z = np.zeros((L1 + 1, L2 + 1), dtype=np.uint8)
    if zero_idxs is not None:
        for idx in zero_idxs:
            z[idx] = 1
    lib().oracle_init_boundaries(MODES[mode], _p(g1), _p(g2), _p(o), _p(t), L1, L2)
    return o, t, z


def end_cell(mode, o):
    """praline/component/align.py:401-431 - returns ((y, x, k), score)."""
    cell = np.zeros(3, dtype=np.int64)
    score = ctypes.c_float(0.0)
    lib().oracle_end_cell(MODES[mode], _p(o), o.shape[0] - 1, o.shape[1] - 1, _p(cell),
                          ctypes.byref(score))
    return tuple(int(c) for c in cell), float(score.value)


def get_paths(t, cell):
    """praline/util/align.py:144-185 - returns one path as int64 [rows, 2]."""
    L1, L2 = t.shape[0] - 1, t.shape[1] - 1
    cell = np.array(cell, dtype=np.int64)
    path = np.zeros((L1 + L2 + 2, 2), dtype=np.int64)
    n = lib().oracle_traceback(_p(t), L1, L2, _p(cell), _p(path))
    if n < 0:
        raise RuntimeError("malformed traceback")
    return path[:n].copy()


def extend_path_semiglobal(path, mat_shape):
    """praline/util/align.py:268-297"""
    path = _c(path, np.int64)
    n, m = mat_shape
    out = np.zeros((n + m, 2), dtype=np.int64)
    k = lib().oracle_extend_path_semiglobal(_p(path), path.shape[0], n - 1, m - 1, _p(out))
    return out[:k].copy()


def raw_pairwise_align(mode, m, g1, g2, zero_idxs=None, return_matrices=False):
    """RawPairwiseAligner.execute (praline/component/align.py:302-447) on raw arrays.

    Returns (score, path[, o, t]).  path is int64 [rows, 2]."""
    m = _c(m, np.float32)
    g1 = _c(g1, np.float32)
    g2 = _c(g2, np.float32)
    o, t, z = init_matrices(mode, g1, g2, zero_idxs)
    _cext_align(mode, m, g1, g2, o, t, z)
    cell, score = end_cell(mode, o)
    path = get_paths(t, cell)
    if mode.startswith("semiglobal"):
        path = extend_path_semiglobal(path, (o.shape[0], o.shape[1]))
    if return_matrices:
        return score, path, o, t
    return score, path


def gap_arrays(L1, L2, gap_series):
    """praline/component/align.py:182-189, 212-217"""
    gs = list(gap_series)
    if len(gs) == 1:
        gs = [gs[0], gs[0]]
    assert len(gs) == 2
    g1 = np.empty((L1, 2), dtype=np.float32)
    g2 = np.empty((L2, 2), dtype=np.float32)
    g1[:, 0], g1[:, 1] = gs[0], gs[1]
    g2[:, 0], g2[:, 1] = gs[0], gs[1]
    return g1, g2


def pairwise_align(mode, profiles_one, profiles_two, score_matrices, gap_series=(-11.0, -1.0),
                   zero_idxs=None, return_matrices=False):
    """PairwiseAligner.execute (praline/component/align.py:88-251) on raw profile arrays:
    one float32 [L, A_t] profile per track set and side, one [A_t, A_t] matrix per set."""
    i1 = [_c(p, np.float32) for p in profiles_one]
    i2 = [_c(p, np.float32) for p in profiles_two]
    s = [_c(x, np.float32) for x in score_matrices]
    i1nz = [build_nonzero_matrix(p) for p in i1]
    i2nz = [build_nonzero_matrix(p) for p in i2]
    m = np.zeros((i1[0].shape[0], i2[0].shape[0]), dtype=np.float32)
    cext_build_scores(i1, i2, i1nz, i2nz, s, m)
    g1, g2 = gap_arrays(m.shape[0], m.shape[1], gap_series)
    res = raw_pairwise_align(mode, m, g1, g2, zero_idxs, return_matrices)
    return res + (m,) if return_matrices else res


def pairwise_score_fast(mode, p1, p2, s, gap_open, gap_extend, rects=None, want_path=False):
    """Single-track-set alignment entirely in C (oracle_pairwise)."""
    p1 = _c(p1, np.float32)
    p2 = _c(p2, np.float32)
    s = _c(s, np.float32)
    L1, L2, A = p1.shape[0], p2.shape[0], p1.shape[1]
    score = ctypes.c_float(0.0)
    r = _c(rects if rects is not None else np.zeros((0, 4)), np.int64).reshape(-1, 4)
    path = np.zeros((L1 + L2 + 2, 2), dtype=np.int64) if want_path else None
    n = lib().oracle_pairwise(MODES[mode], _p(p1), _p(p2), _p(s), A, L1, L2, gap_open, gap_extend,
                              _p(r) if r.shape[0] else None, r.shape[0], ctypes.byref(score),
                              _p(path) if want_path else None)
    if n < 0:
        raise RuntimeError("oracle_pairwise failed")
    return (float(score.value), path[:n].copy()) if want_path else float(score.value)


def batch_scores(mode, arena, row_off, lens, s, pairs, gap_open, gap_extend, threads=1):
    """CPU baseline driver: score-only all-pairs over an arena (see praline_oracle.c)."""
    arena = _c(arena, np.float32)
    row_off = _c(row_off, np.int64)
    lens = _c(lens, np.int32)
    s = _c(s, np.float32)
    pairs = _c(pairs, np.int32).reshape(-1, 2)
    scores = np.zeros(pairs.shape[0], dtype=np.float32)
    err = lib().oracle_batch_scores(MODES[mode], _p(arena), _p(row_off), _p(lens), _p(s),
                                    arena.shape[1], _p(pairs), pairs.shape[0], gap_open,
                                    gap_extend, int(threads), _p(scores))
    if err:
        raise RuntimeError("oracle_batch_scores failed")
    return scores


def batch_align(modes, arena, row_off, lens, s, pairs, gap_open, gap_extend, rects=None, threads=1):
    """Alignments with paths for a pair list in several modes, the match scores of each pair built ONCE in
    the reference's order (parity tests at BASELINE sizes).  arena: concatenated float32 profiles [sum L][A];
    rects: optional list (one entry per pair) of (y0, y1, x0, x1) zero rectangles.
    Returns (scores float32 [n_pairs][n_modes], paths): paths[p][k] is an int32 [rows, 2] array."""
    arena = _c(arena, np.float32)
    row_off = _c(row_off, np.int64)
    lens = _c(lens, np.int32)
    s = _c(s, np.float32)
    pairs = _c(pairs, np.int32).reshape(-1, 2)
    mode_ids = _c([MODES[m] for m in modes], np.int32)
    n, nm = pairs.shape[0], len(mode_ids)
    cap = (lens[pairs[:, 0]].astype(np.int64) + lens[pairs[:, 1]] + 2)
    slot = np.concatenate([[0], np.cumsum(np.repeat(cap, nm))]).astype(np.int64)
    buf = np.zeros((max(int(slot[-1]), 1), 2), dtype=np.int32)
    rows = np.zeros(n * nm, dtype=np.int32)
    scores = np.zeros((n, nm), dtype=np.float32)
    ro = rv = None
    if rects is not None:
        ro = np.zeros(n + 1, dtype=np.int64)
        flat = []
        for p_, rl in enumerate(rects):
            flat.extend(rl)
            ro[p_ + 1] = ro[p_] + len(rl)
        rv = _c(np.array(flat, dtype=np.int64).reshape(-1, 4) if flat else np.zeros((1, 4)), np.int64)
    err = lib().oracle_batch_align(nm, _p(mode_ids), _p(arena), _p(row_off), _p(lens), _p(s), arena.shape[1],
                                   _p(pairs), n, gap_open, gap_extend, _p(ro) if ro is not None else None,
                                   _p(rv) if rv is not None else None, int(threads), _p(scores), _p(buf),
                                   _p(slot), _p(rows))
    if err:
        raise RuntimeError("oracle_batch_align failed")
    paths = [[buf[slot[p_ * nm + k]:slot[p_ * nm + k] + rows[p_ * nm + k]] for k in range(nm)] for p_ in range(n)]
    return scores, paths
