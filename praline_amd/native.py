"""ctypes binding of libpraline_dp.so (include/praline_dp.h) - the only way Python reaches HIP.

Nothing in here computes on the CPU: if the shared library is missing, or no HIP device is
usable, the calls raise NativeError.  The numpy-level helpers mirror the call shapes of the
reference's native module (praline/util/cext.c:506-520 as bound in
praline/component/align.py:18-20) so the operator layer reads like the reference's.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRALINE_LIB") or os.path.join(_HERE, "libpraline_dp.so")

MODES = {"global": 0, "local": 1, "semiglobal_both": 2, "semiglobal_one": 3,
         "semiglobal_two": 4}

OK, ERR_ARG, ERR_DEVICE, ERR_NOMEM, ERR_UNSUPPORTED = 0, -1, -2, -3, -4


MAX_RECTS = 4   # zero rectangles per pair the split-strip kernels hold in registers (PRALINE_MAX_RECTS); plans with more
                # per pair read per-row mask words (k_build_zmask; no limit)


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super(NativeError, self).__init__("libpraline_dp error %d: %s" % (code, message))
        self.code = code


class PralineArray(ctypes.Structure):
    """struct praline_array (pointer + dims + byte strides)."""
    _fields_ = [("data", ctypes.c_void_p), ("dim", ctypes.c_int64 * 3),
                ("stride", ctypes.c_int64 * 3)]


_lib = None


def lib():
    """Load libpraline_dp.so (built by __graft_entry__.build / make -C praline_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(ERR_DEVICE, "%s not found - build it with `make -C praline_amd/csrc` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    pa = ctypes.POINTER(PralineArray)
    L.praline_abi_version.restype = i32
    L.praline_device_count.argtypes = [ctypes.POINTER(i32)]
    L.praline_init.argtypes = [i32]
    L.praline_last_error.restype = ctypes.c_char_p
    L.praline_stream.restype = vp
    L.praline_pool_cached_bytes.restype = i64
    L.praline_build_scores.argtypes = [i32, pa, pa, pa, pa, pa, pa]
    for name in ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"):
        getattr(L, "praline_align_" + name).argtypes = [pa, pa, pa, pa, pa, pa]
    L.praline_align.argtypes = [i32, pa, pa, pa, pa, pa, pa]
    L.praline_raw_align.argtypes = [i32, pa, pa, pa, pa, ctypes.POINTER(f32), vp,
                                    ctypes.POINTER(i64)]
    L.praline_arena_create.argtypes = [i64, vp, i32, vp, vp, ctypes.POINTER(vp)]
    if hasattr(L, "praline_arena_begin"):
        L.praline_arena_begin.argtypes = [i64, vp, i32, ctypes.POINTER(vp)]
        L.praline_arena_put_rows.argtypes = [vp, i64, i64, vp]
        L.praline_arena_finish.argtypes = [vp, vp]
    if hasattr(L, "praline_host_alloc"):
        L.praline_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
        L.praline_host_free.argtypes = [vp]
    L.praline_arena_destroy.argtypes = [vp]
    L.praline_arena_set_track_sets.argtypes = [vp, i32, vp]
    L.praline_arena_set_counts.argtypes = [vp, vp, i64, i64]
    L.praline_arena_append_merged.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.praline_arena_append_merged_many.argtypes = [vp, vp, i64, vp, vp, vp]
    L.praline_set_match_mode.argtypes = [i32]
    L.praline_arena_premultiply.argtypes = [vp]
    L.praline_plan_create.argtypes = [vp, i64, vp, i32, vp, vp, ctypes.POINTER(vp)]
    L.praline_plan_destroy.argtypes = [vp]
    if hasattr(L, "praline_sched_prepare"):
        L.praline_sched_prepare.argtypes = [i64, vp, i64, vp, ctypes.POINTER(vp)]
        L.praline_sched_destroy.argtypes = [vp]
        L.praline_plan_create_prepared.argtypes = [vp, i64, vp, vp, ctypes.POINTER(vp)]
    L.praline_plan_cells.argtypes = [vp]
    L.praline_plan_cells.restype = i64
    L.praline_arena_counts_reset.argtypes = [vp]
    L.praline_arena_counts_bind.argtypes = [vp, vp]
    L.praline_plan_add_counts.argtypes = [vp, ctypes.c_int, f32, ctypes.c_int]
    L.praline_arena_counts_read.argtypes = [vp, vp]
    L.praline_plan_path_bounds.argtypes = [vp, vp]
    L.praline_plan_mask_path_bounds.argtypes = [vp]
    for name in ("praline_arena_counts_reset", "praline_plan_add_counts", "praline_arena_counts_read",
                 "praline_plan_path_bounds"):
        getattr(L, name).restype = ctypes.c_int
    for name in ("praline_plan_steps", "praline_plan_tasks"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i64
    L.praline_plan_path_capacity.argtypes = [vp]
    L.praline_plan_path_capacity.restype = i64
    L.praline_plan_run.argtypes = [vp, i32, f32, f32, vp]
    L.praline_plan_run_gaps.argtypes = [vp, i32, vp]
    L.praline_arena_set_gap_scores.argtypes = [vp, vp]
    L.praline_plan_scores.argtypes = [vp, vp]
    L.praline_plan_device_scores.argtypes = [vp]
    L.praline_plan_device_scores.restype = vp
    L.praline_plan_paths.argtypes = [vp, vp, vp, vp]
    L.praline_batch_scores.argtypes = [vp, i32, f32, f32, i64, vp, vp]
    L.praline_plan_last_timing.argtypes = [vp, ctypes.POINTER(f32)]
    L.praline_plan_kernel_resources.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.praline_arena_match_scores.argtypes = [vp, i32, i32, i32, vp]
    L.praline_arena_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                     ctypes.POINTER(i32)]
    L.praline_plan_match_kind.argtypes = [vp]
    L.praline_plan_tile_producer.argtypes = [vp]
    L.praline_merge_order.argtypes = [i64, vp, i32, vp]
    L.praline_plan_kernel_name.argtypes = [vp, ctypes.c_char_p, i64]
    L.praline_raw_batch_create.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.praline_raw_batch_create_v.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.praline_raw_batch_run.argtypes = [vp, vp, i32]
    L.praline_raw_batch_results.argtypes = [vp, vp, vp]
    L.praline_raw_batch_paths.argtypes = [vp, vp, i64]
    L.praline_raw_batch_cells.argtypes = [vp]
    L.praline_raw_batch_cells.restype = i64
    L.praline_raw_batch_last_timing.argtypes = [vp, ctypes.POINTER(f32)]
    L.praline_raw_batch_destroy.argtypes = [vp]
    L.praline_raw_batch_destroy.restype = None
    if L.praline_abi_version() != 1:
        raise NativeError(ERR_ARG, "ABI version mismatch")
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise NativeError(rc, lib().praline_last_error().decode("utf-8", "replace"))


def device_count():
    n = ctypes.c_int(0)
    rc = lib().praline_device_count(ctypes.byref(n))
    return n.value if rc == OK else 0


def keep_host_memory_mapped(enable=True):
    """Stage-sized host buffers (pair lists, count arenas, the plans' schedules: 4-30 MB each) come and go with every
    submission; glibc serves blocks of that size with mmap and hands them back with munmap, so every one of them is
    page-faulted in again - about 1 ms per MB on the MI355X hosts, 30-40 ms of a 75 ms preprofile stage on C3.  This
    raises glibc's mmap and trim thresholds (mallopt: blocks up to 32 MB from the heap, the heap not trimmed) so that
    such buffers are recycled by malloc.  Process-wide, which is why it is a switch: init() applies it unless
    PRALINE_KEEP_HOST_MEMORY=0."""
    try:
        libc = ctypes.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_MMAP_THRESHOLD = -1, -3
        if enable:
            return bool(libc.mallopt(M_MMAP_THRESHOLD, 32 << 20)) and bool(libc.mallopt(M_TRIM_THRESHOLD, (1 << 31) - 1))
        return bool(libc.mallopt(M_MMAP_THRESHOLD, 128 << 10)) and bool(libc.mallopt(M_TRIM_THRESHOLD, 128 << 10))
    except (OSError, AttributeError):
        return False


def init(device=0):
    _check(lib().praline_init(int(device)))
    if os.environ.get("PRALINE_KEEP_HOST_MEMORY", "1") != "0":
        keep_host_memory_mapped(True)


def synchronize():
    _check(lib().praline_synchronize())


def stream_handle():
    return lib().praline_stream()


MATCH_MODES = {"fast": 0, "f32": 1, "ref": 2, None: -1}


def set_match_mode(kind):
    """How plans created from now on evaluate the match scores (praline_set_match_mode): "fast" (matrix pipe, f16
    hi/lo split), "f32" (fp32 MFMA chain), "ref" (the reference's own summation order on the VALU: scores and
    alignments bit-identical to the reference for any profiles, ~10-20x slower); None = the PRALINE_MM default."""
    _check(lib().praline_set_match_mode(MATCH_MODES[kind]))


def get_match_mode():
    return {v: k for k, v in MATCH_MODES.items()}[int(lib().praline_get_match_mode())]


def merge_order(dist, linkage):
    """Clustering merge order on the host side of the library (praline_merge_order, csrc/cluster.cpp)."""
    d = np.ascontiguousarray(dist, dtype=np.float64)
    n = d.shape[0]
    out = np.zeros((max(n - 1, 0), 2), dtype=np.int32)
    rc = lib().praline_merge_order(n, d.ctypes.data, {'single': 0, 'complete': 1, 'average': 2}[linkage], out.ctypes.data)
    if rc != OK:
        raise NativeError(rc, "praline_merge_order: bad arguments")
    return [tuple(int(v) for v in row) for row in out]


def pool_trim():
    """Return the library's cached device blocks to the driver (praline_pool_trim)."""
    _check(lib().praline_pool_trim())


def pool_cached_bytes():
    return int(lib().praline_pool_cached_bytes())


def shutdown():
    """Release the library's stream and every cached device buffer (praline_shutdown)."""
    _check(lib().praline_shutdown())


def _arr(a):
    """numpy array (any strides) -> praline_array."""
    pa = PralineArray()
    pa.data = a.ctypes.data
    for k in range(3):
        pa.dim[k] = a.shape[k] if k < a.ndim else 1
        pa.stride[k] = a.strides[k] if k < a.ndim else 0
    return pa


def _arr_list(arrs):
    return (PralineArray * len(arrs))(*[_arr(a) for a in arrs])


def _need(a, dtype, name):
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise TypeError("%s must be a numpy array of dtype %s" % (name, np.dtype(dtype)))
    return a


# ---- drop-in twins of the reference's native functions -----------------------------------------
def cext_build_scores(i1s, i2s, i1nzs, i2nzs, ss, m):
    """cext_build_scores(i1s, i2s, i1nzs, i2nzs, ss, m) -> None   (praline/util/cext.c:308-455)

    Writes the match-score matrix into m in place; arbitrary strides are honoured.  The nonzero
    index lists are accepted but unused (dense MFMA contraction on the device)."""
    n = len(i1s)
    for k in range(n):
        _need(i1s[k], np.float32, "i1s[%d]" % k)
        _need(i2s[k], np.float32, "i2s[%d]" % k)
        _need(ss[k], np.float32, "ss[%d]" % k)
    _need(m, np.float32, "m")
    pm = _arr(m)
    _check(lib().praline_build_scores(n, _arr_list(i1s), _arr_list(i2s), None, None,
                                      _arr_list(ss), ctypes.byref(pm)))


def _cext_align(mode, m, g1, g2, o, t, z):
    args = [_arr(_need(m, np.float32, "m")), _arr(_need(g1, np.float32, "g1")),
            _arr(_need(g2, np.float32, "g2")), _arr(_need(o, np.float32, "o")),
            _arr(_need(t, np.uint8, "t")), _arr(_need(z, np.uint8, "z"))]
    fn = getattr(lib(), "praline_align_" + mode)
    _check(fn(*[ctypes.byref(a) for a in args]))


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


def raw_align(mode, m, g1, g2, z=None):
    """RawPairwiseAligner's numeric core on the device (praline/component/align.py:357-447):
    returns (score, path int32 [rows, 2]) without moving o / t back to the host."""
    m = _need(m, np.float32, "m")
    g1 = _need(g1, np.float32, "g1")
    g2 = _need(g2, np.float32, "g2")
    L1, L2 = m.shape
    path = np.zeros((L1 + L2 + 2, 2), dtype=np.int32)
    score = ctypes.c_float(0.0)
    rows = ctypes.c_int64(0)
    am, ag1, ag2 = _arr(m), _arr(g1), _arr(g2)
    az = _arr(_need(z, np.uint8, "z")) if z is not None else None
    _check(lib().praline_raw_align(MODES[mode], ctypes.byref(am), ctypes.byref(ag1),
                                   ctypes.byref(ag2), ctypes.byref(az) if az is not None else None,
                                   ctypes.byref(score), path.ctypes.data, ctypes.byref(rows)))
    return float(score.value), path[:rows.value].copy()


class RawBatch(object):
    """A list of RawPairwiseAligner requests on the device (praline_raw_batch_*; praline/component/align.py:254-447):
    `requests` = (m, g1, g2, zero_idxs or None) per request - the reference operator's `match_score_model.scores`,
    `gap_score_model_one / two.scores` and `zero_idxs` inputs.  run(modes) aligns all of them in one launch; results()
    returns (scores float32 [n], [path int32 [rows, 2] per request])."""

    def __init__(self, requests):
        n = len(requests)
        if n == 0:
            raise ValueError("empty request list")
        ms, g1s, g2s, zs = [], [], [], []       # (the arrays themselves: they are handed over by pointer, one per request)
        l1 = np.zeros(n, np.int32)
        l2 = np.zeros(n, np.int32)
        zoff = np.zeros(n + 1, np.int64)
        for r, (m, g1, g2, zero_idxs) in enumerate(requests):
            m = np.ascontiguousarray(m, dtype=np.float32)
            g1 = np.ascontiguousarray(g1, dtype=np.float32)
            g2 = np.ascontiguousarray(g2, dtype=np.float32)
            if m.ndim != 2 or g1.shape != (m.shape[0], 2) or g2.shape != (m.shape[1], 2):
                raise ValueError("request %d: m %s, g1 %s, g2 %s do not fit" % (r, m.shape, g1.shape, g2.shape))
            l1[r], l2[r] = m.shape
            ms.append(m); g1s.append(g1); g2s.append(g2)
            if zero_idxs is not None and len(zero_idxs):
                z = np.asarray(zero_idxs, dtype=np.int64).reshape(-1, 2)
                # (the reference indexes a numpy array with these tuples: negative indices count from the end)
                z = np.where(z < 0, z + np.array([m.shape[0] + 1, m.shape[1] + 1]), z)
                if (z < 0).any() or (z[:, 0] > m.shape[0]).any() or (z[:, 1] > m.shape[1]).any():
                    raise IndexError("request %d: zero_idxs outside the %d x %d matrix" % (r, m.shape[0] + 1, m.shape[1] + 1))
                zs.append(z.astype(np.int32))
                zoff[r + 1] = zoff[r] + len(z)
            else:
                zoff[r + 1] = zoff[r]
        self.n, self.l1, self.l2 = n, l1, l2
        pm = np.array([a.ctypes.data for a in ms], dtype=np.uint64)
        pg1 = np.array([a.ctypes.data for a in g1s], dtype=np.uint64)
        pg2 = np.array([a.ctypes.data for a in g2s], dtype=np.uint64)
        z_all = np.ascontiguousarray(np.concatenate(zs)) if zs else None
        h = ctypes.c_void_p()
        _check(lib().praline_raw_batch_create_v(n, l1.ctypes.data, l2.ctypes.data, pm.ctypes.data, pg1.ctypes.data,
                                                pg2.ctypes.data, zoff.ctypes.data if z_all is not None else None,
                                                z_all.ctypes.data if z_all is not None else None, ctypes.byref(h)))
        del ms, g1s, g2s                        # (copied: praline_raw_batch_create_v returns after its uploads)
        self._h = h

    @classmethod
    def from_pointers(cls, l1, l2, m_ptr, g1_ptr, g2_ptr):
        """Requests whose arrays already lie one after the other in host or DEVICE memory (m: the l1[r] x l2[r] float32
        matrices, g1 / g2: float32 [l1[r]][2] / [l2[r]][2]); no zero cells.  The arrays are copied into the batch's own
        layout by the call and may be released afterwards."""
        self = cls.__new__(cls)
        self.l1 = np.ascontiguousarray(l1, dtype=np.int32)
        self.l2 = np.ascontiguousarray(l2, dtype=np.int32)
        self.n = len(self.l1)
        h = ctypes.c_void_p()
        _check(lib().praline_raw_batch_create(self.n, self.l1.ctypes.data, self.l2.ctypes.data, int(m_ptr), int(g1_ptr),
                                              int(g2_ptr), None, None, ctypes.byref(h)))
        self._h = h
        return self

    @property
    def cells(self):
        return int(lib().praline_raw_batch_cells(self._h))

    def run(self, modes):
        """modes: one mode name for every request, or a list of n names.  Asynchronous."""
        if isinstance(modes, str):
            _check(lib().praline_raw_batch_run(self._h, None, MODES[modes]))
        else:
            arr = np.array([MODES[mo] for mo in modes], dtype=np.int32)
            if len(arr) != self.n:
                raise ValueError("%d modes for %d requests" % (len(arr), self.n))
            _check(lib().praline_raw_batch_run(self._h, arr.ctypes.data, 0))
        return self

    def results(self, paths=True):
        scores = np.zeros(self.n, np.float32)
        rows = np.zeros(self.n, np.int64)
        _check(lib().praline_raw_batch_results(self._h, scores.ctypes.data, rows.ctypes.data))
        if not paths:
            return scores, None
        total = int(rows.sum())
        flat = np.zeros((max(total, 1), 2), np.int32)
        _check(lib().praline_raw_batch_paths(self._h, flat.ctypes.data, total))
        ends = np.cumsum(rows)
        return scores, [flat[e - k:e] for e, k in zip(ends, rows)]

    def last_kernel_ms(self):
        ms = ctypes.c_float(0.0)
        _check(lib().praline_raw_batch_last_timing(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def close(self):
        if self._h:
            lib().praline_raw_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- batched path --------------------------------------------------------------------------------
# Page-locked staging for the profiles of an arena (praline_host_alloc): the list of per-sequence arrays is concatenated
# straight into it and goes up by DMA.  One buffer per process, grown on demand, reused by every Arena (the upload has
# finished when praline_arena_create returns); lists beyond the cap take the pageable path.
_STAGE_CAP = 1 << 30
_PARTS_MIN_BYTES = 4 << 20   # arenas from this size on are uploaded in two parts (Arena._create_in_parts)
_stage = {"ptr": None, "view": None}


def _stage_view(n_floats):
    if n_floats * 4 > _STAGE_CAP or not hasattr(lib(), "praline_host_alloc"):
        return None
    v = _stage["view"]
    if v is None or v.size < n_floats:
        if _stage["ptr"] is not None:
            _stage["view"] = None
            lib().praline_host_free(_stage["ptr"])
            _stage["ptr"] = None
        want = max(1 << 22, 1 << int(n_floats - 1).bit_length())
        p = ctypes.c_void_p()
        if lib().praline_host_alloc(want * 4, ctypes.byref(p)) != OK:
            return None
        _stage["ptr"] = p
        v = _stage["view"] = np.ctypeslib.as_array((ctypes.c_float * want).from_address(p.value))
    return v[:n_floats]


_result = {"ptr": None, "view": None}


def _result_view(n_floats):
    """A page-locked float32 buffer for score read-backs (grown on demand, at most 64 MB)."""
    if n_floats * 4 > (64 << 20) or not hasattr(lib(), "praline_host_alloc"):
        return None
    v = _result["view"]
    if v is None or v.size < n_floats:
        if _result["ptr"] is not None:
            _result["view"] = None
            lib().praline_host_free(_result["ptr"])
            _result["ptr"] = None
        want = max(1 << 16, 1 << int(n_floats - 1).bit_length())
        p = ctypes.c_void_p()
        if lib().praline_host_alloc(want * 4, ctypes.byref(p)) != OK:
            return None
        _result["ptr"] = p
        v = _result["view"] = np.ctypeslib.as_array((ctypes.c_float * want).from_address(p.value))
    return v[:n_floats]


_counts_stage = {"view": None}


def _counts_view(n_ints):
    """A reused int32 host buffer for count read-backs (grown on demand, at most 1 GB): its pages are touched once, where a
    fresh 27 MB array per call costs ~20 ms of page faults.  Ordinary memory on purpose - numpy reads the counts back out
    of it, and CPU reads of page-locked (fine-grained, uncached) memory ran at 1 GB/s."""
    if n_ints * 4 > (1 << 30):
        return None
    v = _counts_stage["view"]
    if v is None or v.size < n_ints:
        v = _counts_stage["view"] = np.empty(max(1 << 16, 1 << int(n_ints - 1).bit_length()), dtype=np.int32)
    return v[:n_ints]


def _stage_profiles(profiles, A):
    """The float32 [sum L, A] concatenation of `profiles`, in page-locked memory when they are float32 already."""
    first = profiles[0]
    if isinstance(first, np.ndarray) and first.dtype == np.float32 and first.ndim == 2 and first.shape[1] == A:
        rows = 0
        for p in profiles:
            rows += len(p)
        v = _stage_view(rows * A)
        if v is not None:
            try:
                return np.concatenate(profiles, axis=0, out=v.reshape(rows, A), casting="no")
            except (TypeError, ValueError):
                pass   # (mixed dtypes or shapes: the general path below reports what is wrong)
    return np.ascontiguousarray(np.concatenate(profiles, axis=0), dtype=np.float32)


class Arena(object):
    """Profiles of N sequences resident in HBM (praline_arena_create)."""

    def __init__(self, profiles, score_matrix, set_sizes=None):
        """profiles: list of float32 [L_s, A] arrays; score_matrix: float32 [A, A]; set_sizes: widths of the track
        sets concatenated along the alphabet axis (only the "ref" match mode needs them)."""
        A = int(score_matrix.shape[0])
        self.lens = np.array([p.shape[0] for p in profiles], dtype=np.int32)
        S = np.ascontiguousarray(score_matrix, dtype=np.float32)
        self.n_seqs = len(profiles)
        self.A = A
        h = self._create_in_parts(profiles, A, S)
        if h is None:
            cat = _stage_profiles(profiles, A)
            if cat.shape[1] != A:
                raise ValueError("profile width %d != score matrix size %d" % (cat.shape[1], A))
            h = ctypes.c_void_p()
            _check(lib().praline_arena_create(self.n_seqs, self.lens.ctypes.data, A, cat.ctypes.data,
                                              S.ctypes.data, ctypes.byref(h)))
        self._h = h
        if set_sizes is not None and len(set_sizes) > 1:
            sz = np.ascontiguousarray(set_sizes, dtype=np.int32)
            _check(lib().praline_arena_set_track_sets(h, len(sz), sz.ctypes.data))

    def _create_in_parts(self, profiles, A, S):
        """Large lists of float32 profiles: the concatenation into page-locked staging is done in two or four parts, and
        each part goes up (praline_arena_put_rows, a DMA) while numpy copies the next.  Returns the arena handle, or
        None when this road does not apply (small or mixed inputs, no staging)."""
        rows = int(self.lens.sum())
        if rows * A * 4 < _PARTS_MIN_BYTES or len(profiles) < 8 or not hasattr(lib(), "praline_arena_begin"):
            return None
        if not all(isinstance(p, np.ndarray) and p.dtype == np.float32 and p.ndim == 2 and p.shape[1] == A for p in profiles):
            return None
        v = _stage_view(rows * A)
        if v is None:
            return None
        out = v.reshape(rows, A)
        # (C2's 11 MB, arena creation alone: one part 0.645 ms, two 0.57-0.59, three or four 0.535: scripts/exp_arena_parts.py)
        n_parts = int(os.environ.get("PRALINE_ARENA_PARTS", "4" if rows * A * 4 >= 2 * _PARTS_MIN_BYTES else "2"))
        n_parts = max(1, min(n_parts, len(profiles)))
        cuts = [len(profiles) * k // n_parts for k in range(n_parts + 1)]
        row_at = np.concatenate([[0], np.cumsum(self.lens, dtype=np.int64)])
        h = ctypes.c_void_p()
        _check(lib().praline_arena_begin(self.n_seqs, self.lens.ctypes.data, A, ctypes.byref(h)))
        try:
            for k in range(n_parts):
                r0, r1 = int(row_at[cuts[k]]), int(row_at[cuts[k + 1]])
                part = out[r0:r1]
                np.concatenate(profiles[cuts[k]:cuts[k + 1]], axis=0, out=part, casting="no")
                _check(lib().praline_arena_put_rows(h, r0, r1 - r0, part.ctypes.data))
        except Exception:
            lib().praline_arena_destroy(h)
            raise
        _check(lib().praline_arena_finish(h, S.ctypes.data))   # (destroys the arena when it fails)
        return h

    def set_gap_scores(self, gap_scores):
        """Per-position gap scores (praline_arena_set_gap_scores): a list of float32 [L_s, 2] arrays, one per sequence,
        or one [sum L, 2] array; None removes them.  Plans created afterwards can `run_gaps`."""
        if gap_scores is None:
            _check(lib().praline_arena_set_gap_scores(self._h, None))
            return
        g = gap_scores if isinstance(gap_scores, np.ndarray) else np.concatenate([np.asarray(x, dtype=np.float32) for x in gap_scores], axis=0)
        g = np.ascontiguousarray(g, dtype=np.float32)
        if g.shape != (int(self.lens.sum()), 2):
            raise ValueError("gap scores must have shape (%d, 2)" % int(self.lens.sum()))
        _check(lib().praline_arena_set_gap_scores(self._h, g.ctypes.data))

    def set_counts(self, counts, reserve_seqs=0, reserve_rows=0):
        """The integer counts behind the profile rows, int32 [sum L, A] (praline_arena_set_counts): makes the arena
        growable by append_merged."""
        c = np.ascontiguousarray(counts, dtype=np.int32)
        if c.shape != (int(self.lens.sum()), self.A):
            raise ValueError("counts must have shape (%d, %d)" % (int(self.lens.sum()), self.A))
        _check(lib().praline_arena_set_counts(self._h, c.ctypes.data, int(reserve_seqs), int(reserve_rows)))

    def append_merged(self, plan, pair_index=0):
        """Merge the two sequences of a path plan's pair along its device path into a NEW arena sequence
        (praline_arena_append_merged); returns (index, length)."""
        idx, ln = ctypes.c_int32(0), ctypes.c_int32(0)
        _check(lib().praline_arena_append_merged(self._h, plan._h, int(pair_index), ctypes.byref(idx), ctypes.byref(ln)))
        self.lens = np.append(self.lens, np.int32(ln.value))
        self.n_seqs += 1
        return int(idx.value), int(ln.value)

    def append_merged_many(self, plan, pair_indices):
        """append_merged for several pairs of one plan (praline_arena_append_merged_many): list of (index, length)."""
        pi = np.ascontiguousarray(pair_indices, dtype=np.int64)
        n = int(pi.shape[0])
        idx = np.zeros(n, dtype=np.int32)
        ln = np.zeros(n, dtype=np.int32)
        _check(lib().praline_arena_append_merged_many(self._h, plan._h, n, pi.ctypes.data, idx.ctypes.data, ln.ctypes.data))
        self.lens = np.append(self.lens, ln)
        self.n_seqs += n
        return [(int(i), int(l)) for i, l in zip(idx, ln)]

    def premultiply(self):
        _check(lib().praline_arena_premultiply(self._h))

    def match_scores(self, one, two, kind=0):
        """Dense match-score matrix of a pair exactly as the kernels evaluate it (diagnostics):
        kind 0 = fp32 MFMA chain, kind 1 = f16 hi/lo split on the matrix pipe, kind 2 = reference order."""
        m = np.zeros((int(self.lens[one]), int(self.lens[two])), dtype=np.float32)
        _check(lib().praline_arena_match_scores(self._h, int(one), int(two), int(kind), m.ctypes.data))
        return m

    def counts_bind(self, device_ptr):
        """Accumulate the preprofile counts in a caller-owned DEVICE buffer int32 [sum L][A] (praline_arena_counts_bind);
        0 / None goes back to the arena's own buffer."""
        _check(lib().praline_arena_counts_bind(self._h, ctypes.c_void_p(device_ptr or 0)))

    def counts_reset(self):
        """Zero the preprofile count buffer int32 [sum L, A] (praline_arena_counts_reset)."""
        _check(lib().praline_arena_counts_reset(self._h))

    def counts(self, staged=False):
        """The preprofile counts accumulated by Plan.add_counts, int32 [sum L, A] on the host.  staged: a view of the
        process's reused read-back buffer (no page faults of a fresh 27 MB array on C3) - valid until the next staged
        read-back; the caller copies what it keeps."""
        rows = int(self.lens.sum())
        if staged:
            v = _counts_view(rows * self.A)
            if v is not None:
                _check(lib().praline_arena_counts_read(self._h, v.ctypes.data))
                return v.reshape(rows, self.A)
        out = np.empty((rows, self.A), dtype=np.int32)
        _check(lib().praline_arena_counts_read(self._h, out.ctypes.data))
        return out

    def info(self):
        vals = [ctypes.c_int(0) for _ in range(4)]
        _check(lib().praline_arena_info(self._h, *[ctypes.byref(v) for v in vals]))
        return dict(zip(("n_active", "mfma_steps_f32", "f16_ranges", "f16_terms"), [v.value for v in vals]))

    def close(self):
        if getattr(self, "_h", None):
            lib().praline_arena_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedSchedule(object):
    """The host scheduling of a scores-only plan, computed from the sequence lengths and the pair list alone
    (praline_sched_prepare) - before, or on another thread beside, the creation of the arena."""

    def __init__(self, lens, pairs):
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        self.pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        h = ctypes.c_void_p()
        _check(lib().praline_sched_prepare(len(self.lens), self.lens.ctypes.data, self.pairs.shape[0], self.pairs.ctypes.data,
                                           ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().praline_sched_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_sched_pool = None


def prepare_schedule_async(lens, pairs):
    """PreparedSchedule(lens, pairs) on a worker thread (the call spends its time in the library, outside the
    interpreter lock): start it, create the Arena of the same sequences, then Plan(arena, pairs, prepared=future)."""
    global _sched_pool
    if _sched_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _sched_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="praline-sched")
    lib()   # (loaded on the calling thread)
    return _sched_pool.submit(PreparedSchedule, lens, pairs)


class Plan(object):
    """A scheduled pair list (praline_plan_create)."""

    def __init__(self, arena, pairs, want_paths=False, rects=None, prepared=None):
        """pairs: int [n, 2] (sequence_one, sequence_two); rects: optional list (one entry per
        pair) of lists of (y0, y1, x0, x1) inclusive zero rectangles, or an int array [n, k, 4];
        prepared: a PreparedSchedule of the same lengths and pairs (or the future of one) for a scores-only plan."""
        self.arena = arena
        if prepared is not None and hasattr(prepared, "result"):
            prepared = prepared.result()
        if prepared is not None and (want_paths or rects is not None):
            raise ValueError("a prepared schedule serves scores-only plans")
        self.pairs = prepared.pairs if prepared is not None and pairs is prepared.pairs else \
            np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        self.n = self.pairs.shape[0]
        self.want_paths = bool(want_paths)
        ro = rv = None
        if isinstance(rects, np.ndarray):
            # packed form: int [n, k, 4], the same k rectangles for every pair (k Waterman-Eggert passes)
            rv = np.ascontiguousarray(rects, dtype=np.int32).reshape(self.n, -1, 4)
            ro = (np.arange(self.n + 1, dtype=np.int64) * rv.shape[1]).astype(np.int32)
            rv = rv.reshape(-1, 4)
            if rv.shape[0] == 0:
                rv = np.zeros((1, 4), dtype=np.int32)
        elif rects is not None:
            ro = np.zeros(self.n + 1, dtype=np.int32)
            flat = []
            for p, rl in enumerate(rects):
                flat.extend(rl)
                ro[p + 1] = ro[p] + len(rl)
            rv = np.ascontiguousarray(np.array(flat, dtype=np.int32).reshape(-1, 4))
            if rv.shape[0] == 0:
                rv = np.zeros((1, 4), dtype=np.int32)
        self._keep = (ro, rv)
        h = ctypes.c_void_p()
        if prepared is not None:
            _check(lib().praline_plan_create_prepared(arena._h, self.n, self.pairs.ctypes.data, prepared._h, ctypes.byref(h)))
            prepared.close()
        else:
            _check(lib().praline_plan_create(arena._h, self.n, self.pairs.ctypes.data, int(want_paths),
                                             ro.ctypes.data if ro is not None else None,
                                             rv.ctypes.data if rv is not None else None,
                                             ctypes.byref(h)))
        self._h = h
        self.cells = int(lib().praline_plan_cells(h))
        self.steps = int(lib().praline_plan_steps(h))   # wavefront steps per run (1024 cells each, incl. padding)
        self.tasks = int(lib().praline_plan_tasks(h))   # 32-pair tasks

    def run(self, mode, gap_open, gap_extend, d_scores=None):
        """Asynchronous launch on the library stream.  d_scores: optional DEVICE pointer (int)."""
        _check(lib().praline_plan_run(self._h, MODES[mode], float(gap_open), float(gap_extend),
                                      ctypes.c_void_p(d_scores) if d_scores else None))

    def run_gaps(self, mode, d_scores=None):
        """run() with the arena's per-position gap scores (Arena.set_gap_scores; praline_plan_run_gaps)."""
        _check(lib().praline_plan_run_gaps(self._h, MODES[mode], ctypes.c_void_p(d_scores) if d_scores else None))

    def scores(self):
        # (read back into page-locked memory - a DMA the host only waits for - and copied out)
        v = _result_view(self.n) if self.n else None
        if v is not None:
            _check(lib().praline_plan_scores(self._h, v.ctypes.data))
            return v.copy()
        out = np.zeros(self.n, dtype=np.float32)
        _check(lib().praline_plan_scores(self._h, out.ctypes.data))
        return out

    def match_kind(self):
        """0: this plan's run() evaluates match scores with the fp32 MFMA chain, 1: f16 split, 2: reference order."""
        return int(lib().praline_plan_match_kind(self._h))

    def tile_producer(self):
        """0: the fill forms its match scores itself; dense-tile plans: 1 k_match_tile, 2 one thread per cell (both in the
        reference's summation order), 3 the fp32 MFMA chain (praline_plan_tile_producer)."""
        return int(lib().praline_plan_tile_producer(self._h))

    def device_scores_ptr(self):
        return lib().praline_plan_device_scores(self._h)

    def kernel_name(self):
        """The DP kernel instance the last run() launched, as rocprofv3 names it (praline_plan_kernel_name)."""
        buf = ctypes.create_string_buffer(200)
        _check(lib().praline_plan_kernel_name(self._h, buf, 200))
        return buf.value.decode()

    def kernel_ms(self):
        ms = ctypes.c_float(0.0)
        _check(lib().praline_plan_last_timing(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def kernel_resources(self):
        """{vgprs, lds_bytes, waves_per_simd} of the kernel instance the last run launched (zeros when not reported)."""
        v, l, w = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(lib().praline_plan_kernel_resources(self._h, ctypes.byref(v), ctypes.byref(l), ctypes.byref(w)))
        return {"vgprs": int(v.value), "lds_bytes": int(l.value), "waves_per_simd": int(w.value)}

    def add_counts(self, threshold=None, local=False):
        """Fold this plan's (master, slave) paths into the arena's preprofile counts on the device
        (praline_plan_add_counts): compress_path + extend_path_local + merge + get_frequencies."""
        _check(lib().praline_plan_add_counts(self._h, 0 if threshold is None else 1,
                                             0.0 if threshold is None else float(threshold), 1 if local else 0))

    def mask_path_bounds(self):
        """Add every pair's current path bounding box to its zero rectangles ON THE DEVICE (the next Waterman-Eggert
        iteration runs on this same plan; praline_plan_mask_path_bounds)."""
        _check(lib().praline_plan_mask_path_bounds(self._h))

    def path_bounds(self):
        """int32 [n, 4]: (y0, y1, x0, x1) of every path - the next Waterman-Eggert mask rectangle."""
        out = np.zeros((max(self.n, 1), 4), dtype=np.int32)
        _check(lib().praline_plan_path_bounds(self._h, out.ctypes.data))
        return out[:self.n]

    def paths_packed(self):
        """(buf int32 [capacity, 2], off int64 [n], rows int32 [n]): path p is buf[off[p]:off[p] + rows[p]]."""
        cap = int(lib().praline_plan_path_capacity(self._h))
        buf = np.empty((max(cap, 1), 2), dtype=np.int32)
        off = np.zeros(max(self.n, 1), dtype=np.int64)
        rows = np.zeros(max(self.n, 1), dtype=np.int32)
        _check(lib().praline_plan_paths(self._h, buf.ctypes.data, off.ctypes.data, rows.ctypes.data))
        return buf, off, rows

    def paths(self):
        """list of int32 [rows, 2] arrays in pair order (views into one buffer)."""
        buf, off, rows = self.paths_packed()
        off = off.tolist()
        rows = rows.tolist()
        return [buf[off[p]:off[p] + rows[p]] for p in range(self.n)]

    def close(self):
        if getattr(self, "_h", None):
            lib().praline_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
