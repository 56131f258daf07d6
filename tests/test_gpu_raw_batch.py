"""GPU tests of the batched RawPairwiseAligner (praline_raw_batch_*, k_rawb_fill / k_rawb_trace; the reference operator:
praline/component/align.py:254-447): lists of requests that each bring their own match scores, gap scores and zero cells,
aligned in one submission.  Checkers: the reference's own outputs (tests/golden/fill_small.npz, written by the real
reference) and the oracle's restatement of RawPairwiseAligner on the same inputs - scores and paths bit for bit."""
import ctypes

import numpy as np
import pytest

from conftest import MODES, load_golden
from oracle import oracle as orc
from praline_amd import component as comp
from praline_amd import container as ct
from praline_amd import core

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    return native


@pytest.fixture(scope="module")
def managers(nat):
    idx = core.TypeIndex()
    idx.autoregister()
    return {"serial": core.Manager(idx), "batch": comp.BatchManager(idx)}


def rand_request(rng, lo, hi, zero_share=0.35, integer=False):
    L1, L2 = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
    if integer:
        m = rng.integers(-4, 12, (L1, L2)).astype(np.float32)     # (many exact ties: every tie flag matters)
    else:
        m = (rng.standard_normal((L1, L2)) * 3 - 0.5).astype(np.float32)
    g1 = np.stack([-rng.uniform(5, 12, L1), -rng.uniform(0.5, 2, L1)], axis=1).astype(np.float32)
    g2 = np.stack([-rng.uniform(5, 12, L2), -rng.uniform(0.5, 2, L2)], axis=1).astype(np.float32)
    if integer:
        g1, g2 = np.round(g1), np.round(g2)
    z = None
    if rng.random() < zero_share:
        z = [(int(rng.integers(0, L1 + 1)), int(rng.integers(0, L2 + 1))) for _ in range(int(rng.integers(1, 60)))]
    return m, g1, g2, z


def dummy_sequence(name, n):
    return ct.Sequence(name, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=np.zeros(n, int)))])


def test_500_raw_requests_in_one_execution(managers, nat, monkeypatch):
    """500 random (m, g1, g2, zero_idxs) requests, the five modes mixed, as ONE Execution under BatchManager: one
    submission (one native.RawBatch, one run), scores and paths equal to the oracle's RawPairwiseAligner."""
    rng = np.random.default_rng(2024)
    made, runs = [], []
    orig_init, orig_run = nat.RawBatch.__init__, nat.RawBatch.run
    monkeypatch.setattr(nat.RawBatch, "__init__", lambda self, *a, **k: (made.append(1), orig_init(self, *a, **k))[1])
    monkeypatch.setattr(nat.RawBatch, "run", lambda self, *a, **k: (runs.append(1), orig_run(self, *a, **k))[1])
    reqs = [rand_request(rng, 1, 400, integer=(k % 2 == 0)) for k in range(500)]
    ex = core.Execution(managers["batch"], "root")
    for k, (m, g1, g2, z) in enumerate(reqs):
        a, b = dummy_sequence("a%d" % k, m.shape[0]), dummy_sequence("b%d" % k, m.shape[1])
        ex.add_task(comp.RawPairwiseAligner).environment(core.Environment({}), core.Environment({})).inputs(
            mode=MODES[k % 5], sequence_one=a, sequence_two=b, match_score_model=ct.MatchScoreModel(a, b, m),
            gap_score_model_one=ct.GapScoreModel(a, g1), gap_score_model_two=ct.GapScoreModel(b, g2), zero_idxs=z)
    outs = core.run(ex)
    assert len(made) == 1 and len(runs) == 1
    for k, ((m, g1, g2, z), out) in enumerate(zip(reqs, outs)):
        s, p = orc.raw_pairwise_align(MODES[k % 5], m, g1, g2, z)
        assert isinstance(out['score'], float) and out['score'] == float(np.float32(s)), (k, MODES[k % 5], m.shape)
        assert np.array_equal(np.asarray(out['alignment'].path), np.asarray(p)), (k, MODES[k % 5], m.shape)
        path = out['alignment'].path
        assert isinstance(path, np.ndarray) if MODES[k % 5].startswith("semiglobal") else isinstance(path, list)


def test_reference_cases_through_the_batch(managers):
    """The raw cases the real reference wrote (tests/golden/fill_small.npz: its scores and paths) as one request list, and the
    same list under the serial manager (one praline_raw_align per request)."""
    d = load_golden("fill_small.npz")
    n = int(d["n_cases"])
    outs = {}
    for name in ("batch", "serial"):
        ex = core.Execution(managers[name], "root")
        for q in range(n):
            p = "c%03d_" % q
            m = d[p + "m"]
            a, b = dummy_sequence("a", m.shape[0]), dummy_sequence("b", m.shape[1])
            zi = [tuple(int(v) for v in r) for r in d[p + "zero_idxs"]] if (p + "zero_idxs") in d.files else None
            ex.add_task(comp.RawPairwiseAligner).environment(core.Environment({}), core.Environment({})).inputs(
                mode=str(d[p + "mode"]), sequence_one=a, sequence_two=b, match_score_model=ct.MatchScoreModel(a, b, m),
                gap_score_model_one=ct.GapScoreModel(a, d[p + "g1"]), gap_score_model_two=ct.GapScoreModel(b, d[p + "g2"]),
                zero_idxs=zi)
        outs[name] = core.run(ex)
    for q in range(n):
        p = "c%03d_" % q
        for name in ("batch", "serial"):
            assert outs[name][q]['score'] == float(d[p + "score"]), (name, q)
            assert np.array_equal(np.array(outs[name][q]['alignment'].path), d[p + "path"]), (name, q)


def test_every_request_in_every_mode_and_shape_corner(nat):
    """The same requests run five times on one batch, once per mode (the inputs stay on the device); shapes around the strip
    and chunk sizes of the kernel (64 rows per wave, 16 columns per register block), single rows and columns."""
    rng = np.random.default_rng(7)
    shapes = [(1, 1), (1, 70), (70, 1), (2, 2), (63, 15), (64, 16), (65, 17), (64, 64), (128, 33), (129, 48), (17, 300),
              (300, 17), (191, 193), (512, 40), (513, 40), (200, 3), (150, 2)]   # (narrow ones: fewer chunk blocks per strip than a traceback tile is wide)
    reqs = []
    for L1, L2 in shapes:
        m, g1, g2, z = rand_request(rng, 1, 1, zero_share=0.0)
        m = rng.integers(-3, 9, (L1, L2)).astype(np.float32)
        g1 = np.stack([-rng.integers(4, 12, L1), -rng.integers(1, 3, L1)], axis=1).astype(np.float32)
        g2 = np.stack([-rng.integers(4, 12, L2), -rng.integers(1, 3, L2)], axis=1).astype(np.float32)
        z = [(int(rng.integers(-L1 - 1, L1 + 1)), int(rng.integers(-L2 - 1, L2 + 1))) for _ in range(5)] if L1 * L2 > 4 else None
        reqs.append((m, g1, g2, z))
    rb = nat.RawBatch(reqs)
    try:
        for mode in MODES:
            scores, paths = rb.run(mode).results()
            for r, (m, g1, g2, z) in enumerate(reqs):
                s, p = orc.raw_pairwise_align(mode, m, g1, g2, z)
                assert scores[r] == np.float32(s), (mode, m.shape)
                assert np.array_equal(paths[r], np.asarray(p)), (mode, m.shape)
    finally:
        rb.close()


def test_long_requests_take_several_rounds(nat):
    """More than 8 x 64 rows: the waves of a workgroup take several strips each and the last wave hands its row to the first
    through memory; more than 2 600 columns: further than the hand-off rings of a whole round reach."""
    rng = np.random.default_rng(3)
    reqs = [rand_request(rng, 1300, 1400, zero_share=1.0), rand_request(rng, 600, 700), rand_request(rng, 40, 50)]
    m = (rng.standard_normal((530, 3100)) * 3 - 0.5).astype(np.float32)
    g1 = np.tile(np.array([[-9.0, -1.0]], np.float32), (530, 1))
    g2 = np.tile(np.array([[-11.0, -1.5]], np.float32), (3100, 1))
    reqs.append((m, g1, g2, None))
    rb = nat.RawBatch(reqs)
    try:
        for modes in (["local", "global", "semiglobal_both", "semiglobal_one"], ["global", "local", "local", "semiglobal_two"]):
            scores, paths = rb.run(modes).results()
            for r, (m, g1, g2, z) in enumerate(reqs):
                s, p = orc.raw_pairwise_align(modes[r], m, g1, g2, z)
                assert scores[r] == np.float32(s), (modes[r], m.shape)
                assert np.array_equal(paths[r], np.asarray(p)), (modes[r], m.shape)
    finally:
        rb.close()


def test_batch_equals_the_single_request_entry_point(nat):
    """praline_raw_batch_* against praline_raw_align (k_raw_align on the reference's own o / t buffers) request by request."""
    rng = np.random.default_rng(12)
    reqs = [rand_request(rng, 20, 260) for _ in range(40)]
    modes = [MODES[int(rng.integers(0, 5))] for _ in reqs]
    rb = nat.RawBatch(reqs)
    try:
        scores, paths = rb.run(modes).results()
    finally:
        rb.close()
    for r, (m, g1, g2, zi) in enumerate(reqs):
        z = None
        if zi:
            z = np.zeros((m.shape[0] + 1, m.shape[1] + 1), np.uint8)
            for idx in zi:
                z[idx] = 1
        s, p = nat.raw_align(modes[r], m, g1, g2, z)
        assert np.float32(s) == scores[r] and np.array_equal(p, paths[r]), (r, modes[r], m.shape)


def test_requests_resident_in_device_memory(nat):
    """m, g1, g2 of the whole list in DEVICE memory (torch tensors): praline_raw_batch_create takes the pointers as they are."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    rng = np.random.default_rng(21)
    reqs = [rand_request(rng, 30, 200, zero_share=0.0) for _ in range(24)]
    dev = torch.device("cuda", 0)
    m = torch.from_numpy(np.concatenate([r[0].reshape(-1) for r in reqs])).to(dev)
    g1 = torch.from_numpy(np.concatenate([r[1].reshape(-1) for r in reqs])).to(dev)
    g2 = torch.from_numpy(np.concatenate([r[2].reshape(-1) for r in reqs])).to(dev)
    torch.cuda.synchronize()
    rb = nat.RawBatch.from_pointers([r[0].shape[0] for r in reqs], [r[0].shape[1] for r in reqs], m.data_ptr(), g1.data_ptr(),
                                    g2.data_ptr())
    del m, g1, g2
    try:
        for mode in ("global", "local", "semiglobal_one"):
            scores, paths = rb.run(mode).results()
            for r, (mm, a, b, _) in enumerate(reqs):
                s, p = orc.raw_pairwise_align(mode, mm, a, b, None)
                assert scores[r] == np.float32(s) and np.array_equal(paths[r], np.asarray(p)), (mode, r)
    finally:
        rb.close()


def test_raw_batch_rejects_misuse(nat):
    L = nat.lib()
    h = ctypes.c_void_p()
    l1 = np.array([3, 4], np.int32); l2 = np.array([5, 2], np.int32)
    m = np.zeros(3 * 5 + 4 * 2, np.float32); g1 = np.zeros(7 * 2, np.float32); g2 = np.zeros(7 * 2, np.float32)
    args = lambda: (2, l1.ctypes.data, l2.ctypes.data, m.ctypes.data, g1.ctypes.data, g2.ctypes.data, None, None, ctypes.byref(h))
    assert L.praline_raw_batch_create(0, *args()[1:]) == nat.ERR_ARG
    assert L.praline_raw_batch_create(2, None, *args()[2:]) == nat.ERR_ARG
    bad = np.array([3, 0], np.int32)
    assert L.praline_raw_batch_create(2, bad.ctypes.data, *args()[2:]) == nat.ERR_ARG and b"shape" in L.praline_last_error()
    assert L.praline_raw_batch_create(*args()) == 0
    try:
        scores = np.zeros(2, np.float32)
        assert L.praline_raw_batch_results(h, scores.ctypes.data, None) == nat.ERR_ARG        # (before any run)
        assert L.praline_raw_batch_run(h, None, 7) == nat.ERR_ARG
        assert L.praline_raw_batch_run(h, None, 0) == 0
        out = np.zeros((4, 2), np.int32)
        assert L.praline_raw_batch_paths(h, out.ctypes.data, 4) == nat.ERR_ARG                 # (before the results)
        rows = np.zeros(2, np.int64)
        assert L.praline_raw_batch_results(h, scores.ctypes.data, rows.ctypes.data) == 0
        assert L.praline_raw_batch_paths(h, out.ctypes.data, 1) == nat.ERR_ARG                 # (too small)
        assert L.praline_raw_batch_cells(h) == 23
    finally:
        L.praline_raw_batch_destroy(h)
    ptrs = np.array([m.ctypes.data, 0], dtype=np.uint64)      # (one pointer per request: a NULL among them)
    good = np.array([g1.ctypes.data, g1.ctypes.data], dtype=np.uint64)
    assert L.praline_raw_batch_create_v(2, l1.ctypes.data, l2.ctypes.data, ptrs.ctypes.data, good.ctypes.data, good.ctypes.data, None, None,
                                        ctypes.byref(h)) == nat.ERR_ARG
    assert L.praline_raw_batch_create_v(2, l1.ctypes.data, l2.ctypes.data, None, good.ctypes.data, good.ctypes.data, None, None,
                                        ctypes.byref(h)) == nat.ERR_ARG
    with pytest.raises(ValueError):
        nat.RawBatch([(np.zeros((3, 4), np.float32), np.zeros((3, 2), np.float32), np.zeros((5, 2), np.float32), None)])
    with pytest.raises(IndexError):
        nat.RawBatch([(np.zeros((3, 4), np.float32), np.zeros((3, 2), np.float32), np.zeros((4, 2), np.float32), [(4, 1)])])
