"""Per-position gap scores in the batched kernels (praline_arena_set_gap_scores / praline_plan_run_gaps): the reference's
fill reads one (open, extend) per position of each sequence (cext.c:155-158,172-175; boundary cells align.py:371-385).
Checker: the oracle's RawPairwiseAligner restatement on the same match scores."""
import numpy as np
import pytest

from conftest import MODES, one_hot, synth_profile
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    yield native
    native.set_match_mode(None)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_gaps(rng, L, exact):
    """(open, extend) rows, all negative; exact: multiples of 1/4 (every DP value stays exact in float32)"""
    if exact:
        g = np.stack([-rng.integers(8, 60, L) / 4.0, -rng.integers(1, 12, L) / 4.0], axis=1)
    else:
        g = np.stack([-rng.uniform(2.0, 15.0, L), -rng.uniform(0.1, 3.0, L)], axis=1)
    return g.astype(np.float32)


def zero_cells(rects, L1, L2):
    return [(y, x) for (y0, y1, x0, x1) in rects for y in range(y0, y1 + 1) for x in range(x0, x1 + 1) if y <= L1 and x <= L2]


@pytest.mark.parametrize("kind", ["onehot", "float", "float-ref"])
def test_per_position_gap_scores_plans(nat, bba, kind):
    """Plans created on an arena with gap scores run the dense-tile instances k_dp_split16_tb<..., PPG>: five modes, scores-only and with paths,
    Waterman-Eggert rectangles in local mode, ragged lengths across the strip boundary; scores bit-identical and paths
    identical to the oracle (match scores: exact for sequences, the fp32 MFMA chain / the reference order for float
    profiles).  A constant-gap run of the same plan still equals the constant-gap oracle."""
    rng = np.random.default_rng({"onehot": 5, "float": 6, "float-ref": 7}[kind])
    lens = [37, 64, 1, 33, 90, 32, 65]
    S = bba["S"]
    if kind == "onehot":
        profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    else:
        profs = [synth_profile(rng, L)[0] for L in lens]
    gaps = [random_gaps(rng, L, exact=(kind == "onehot")) for L in lens]
    n = len(lens)
    pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
    rects = [[(2, 6, 3, 9), (20, 28, 18, 30)] if k % 2 else [] for k in range(len(pairs))]

    def match(i, j):
        if kind == "float":
            return orc.build_scores_fma([profs[i]], [profs[j]], [S])
        m = np.zeros((lens[i], lens[j]), dtype=np.float32)
        orc.cext_build_scores([profs[i]], [profs[j]], [orc.build_nonzero_matrix(profs[i])], [orc.build_nonzero_matrix(profs[j])], [S], m)
        return m

    ms = {(i, j): match(i, j) for i, j in pairs}
    if kind == "float-ref":
        nat.set_match_mode("ref")
    try:
        arena = nat.Arena(profs, S)
        arena.set_gap_scores(gaps)
        for mode in MODES:
            plan = nat.Plan(arena, pairs, want_paths=True)
            plan.run_gaps(mode)
            sc, paths = plan.scores(), plan.paths()
            # (the dense-tile instance with per-position gap scores; tiles: the fp32 MFMA chain, or the reference order on request)
            assert "k_dp_split16_tb<1, 3" in plan.kernel_name() and plan.kernel_name().endswith(", 4, true, false>"), plan.kernel_name()
            assert plan.tile_producer() == (1 if kind == "float-ref" else 3)
            plan.run(mode, -11.0, -1.0)              # the same plan with one constant pair
            sc_c, paths_c = plan.scores(), plan.paths()
            assert plan.kernel_name().endswith(", 4, false, false>"), plan.kernel_name()
            plan.close()
            plan0 = nat.Plan(arena, pairs, want_paths=False)
            plan0.run_gaps(mode)
            sc0 = plan0.scores()
            assert plan0.kernel_name().endswith(", 4, true, true>"), plan0.kernel_name()     # (the fill without flags)
            plan0.run(mode, -11.0, -1.0)
            sc0_c = plan0.scores()
            assert "k_dp_split16<1, 1" in plan0.kernel_name(), plan0.kernel_name()
            assert np.array_equal(bits(sc0_c), bits(sc_c)), mode
            plan0.close()
            for k, (i, j) in enumerate(pairs):
                s_or, p_or = orc.raw_pairwise_align(mode, ms[(i, j)], gaps[i], gaps[j])
                assert sc[k] == np.float32(s_or), (mode, i, j, sc[k], s_or)
                assert np.array_equal(paths[k], p_or), (mode, i, j)
                assert sc0[k] == np.float32(s_or), (mode, i, j, "scores-only")
                g1, g2 = orc.gap_arrays(lens[i], lens[j], (-11.0, -1.0))
                s_c, p_c = orc.raw_pairwise_align(mode, ms[(i, j)], g1, g2)
                assert sc_c[k] == np.float32(s_c) and np.array_equal(paths_c[k], p_c), (mode, i, j, "constant")
        plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
        plan.run_gaps("local")
        sc, paths = plan.scores(), plan.paths()
        plan.close()
        for k, (i, j) in enumerate(pairs):
            s_or, p_or = orc.raw_pairwise_align("local", ms[(i, j)], gaps[i], gaps[j], zero_idxs=zero_cells(rects[k], lens[i], lens[j]) or None)
            assert sc[k] == np.float32(s_or) and np.array_equal(paths[k], p_or), ("rects", i, j)
        # plans made before the gap scores were set, or after they were removed, refuse to run with them
        arena.set_gap_scores(None)
        plan = nat.Plan(arena, pairs[:4], want_paths=False)
        with pytest.raises(Exception):
            plan.run_gaps("global")
        plan.close()
        arena.close()
    finally:
        nat.set_match_mode(None)


def test_per_position_gap_scores_with_many_rectangles(nat, bba):
    """Per-position gap scores on plans with more rectangles per pair than the register-resident masks hold: the dense-tile
    instance reads the per-row mask words of k_build_zmask.  (Round 3 ran these plans on k_dp_batch, whose <LOCAL, traceback,
    MASK = 2, PPG> instance gave wrong local scores at -O3 - also for pairs without rectangles; that kernel is gone.)  Scores
    and paths against the oracle, local and global."""
    rng = np.random.default_rng(5)
    lens = [60, 75, 48, 66, 90, 170, 159]
    profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    gaps = [random_gaps(rng, L, exact=True) for L in lens]
    n = len(lens)
    pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
    rects = []
    for k, (i, j) in enumerate(pairs):
        r = []
        for _ in range([0, 1, 3, 9][k % 4]):
            y0, x0 = int(rng.integers(1, lens[i])), int(rng.integers(1, lens[j]))
            r.append((y0, min(lens[i], y0 + int(rng.integers(0, 12))), x0, min(lens[j], x0 + int(rng.integers(0, 12)))))
        rects.append(r)
    assert max(len(r) for r in rects) > nat.MAX_RECTS
    arena = nat.Arena(profs, bba["S"])
    arena.set_gap_scores(gaps)
    for mode in ("local", "global"):
        plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
        mk = plan.match_kind()
        plan.run_gaps(mode)
        sc, paths = plan.scores(), plan.paths()
        assert "k_dp_split16_tb<1, 3" in plan.kernel_name() and plan.kernel_name().endswith(", true, false, false, 4, true, false>"), plan.kernel_name()
        assert plan.tile_producer() == 3 and mk == 0
        plan.close()
        for k, (i, j) in enumerate(pairs):
            s_or, p_or = orc.raw_pairwise_align(mode, arena.match_scores(int(i), int(j), mk), gaps[i], gaps[j],
                                                zero_idxs=zero_cells(rects[k], lens[i], lens[j]) or None)
            assert sc[k] == np.float32(s_or) and np.array_equal(paths[k], p_or), (mode, i, j, len(rects[k]))
    arena.close()


def test_pairwise_batch_with_gap_score_models(nat, bba):
    """The host mirror: PairwiseBatch.set_gap_scores hands per-sequence GapScoreModels to the device; sequences without
    one keep the gap series."""
    from praline_amd import component as comp
    from praline_amd import container as ct
    rng = np.random.default_rng(9)
    lens = [40, 71, 18, 55]
    idx = [rng.integers(0, 20, L) for L in lens]
    seqs = [ct.Sequence("s%d" % k, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]) for k, v in enumerate(idx)]
    blosum = ct.blosum62()
    batch = comp.PairwiseBatch([[ct.TRACK_ID_INPUT]], [[ct.TRACK_ID_INPUT]], [blosum], [-11.0, -1.0])
    gaps = {}
    for k in (0, 2, 3):
        gaps[k] = random_gaps(rng, lens[k], exact=True)
        batch.set_gap_scores(seqs[k], ct.GapScoreModel(seqs[k], gaps[k]))
    gaps[1] = np.tile(np.array([[-11.0, -1.0]], dtype=np.float32), (lens[1], 1))
    reqs = [(mode, i, j) for mode in MODES for i in range(4) for j in range(4) if i != j]
    for mode, i, j in reqs:
        batch.add(mode, seqs[i], seqs[j])
    scores, paths = batch.run(want_paths=True)
    S = np.asarray(blosum.matrix, dtype=np.float32)
    for (mode, i, j), s_dev, p_dev in zip(reqs, scores, paths):
        p1, p2 = one_hot(idx[i], S.shape[0]), one_hot(idx[j], S.shape[0])
        m = np.zeros((lens[i], lens[j]), dtype=np.float32)
        orc.cext_build_scores([p1], [p2], [orc.build_nonzero_matrix(p1)], [orc.build_nonzero_matrix(p2)], [S], m)
        s_or, p_or = orc.raw_pairwise_align(mode, m, gaps[i], gaps[j])
        assert np.float32(s_dev) == np.float32(s_or) and np.array_equal(p_dev, p_or), (mode, i, j)


def test_per_position_gap_scores_c2_sample(nat, bba):
    """BASELINE C2's float profiles (256 seqs ~400 aa), 2 048 sampled pairs with per-position gap scores - several launch
    chunks of dense match scores under a small budget: scores-only and path plans agree bit for bit, 24 pairs per mode equal
    the oracle on the device's match scores."""
    import os
    from conftest import synth_lengths
    rng = np.random.default_rng(2)
    N = 256
    lens = synth_lengths(rng, N, 400).astype(np.int32)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    gaps = [random_gaps(rng, int(L), exact=False) for L in lens]
    allp = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
    pairs = allp[np.sort(np.random.default_rng(7).choice(len(allp), 2048, replace=False))]
    arena = nat.Arena(profs, bba["S"])
    arena.set_gap_scores(gaps)
    old = os.environ.get("PRALINE_REF_BUDGET_MB")
    os.environ["PRALINE_REF_BUDGET_MB"] = "256"          # 1.3 GB of dense match scores: six chunks
    try:
        for mode in ("global", "local", "semiglobal_both"):
            plan = nat.Plan(arena, pairs, want_paths=True)
            plan.run_gaps(mode)
            sc, paths, kind = plan.scores().copy(), plan.paths(), plan.match_kind()
            plan.close()
            plan0 = nat.Plan(arena, pairs, want_paths=False)
            plan0.run_gaps(mode)
            assert np.array_equal(bits(plan0.scores()), bits(sc)), mode
            plan0.close()
            for k in np.random.default_rng(11).choice(len(pairs), 24, replace=False):
                i, j = pairs[k]
                s_or, p_or = orc.raw_pairwise_align(mode, arena.match_scores(int(i), int(j), kind), gaps[i], gaps[j])
                assert sc[k] == np.float32(s_or) and np.array_equal(paths[k], p_or), (mode, i, j)
    finally:
        if old is None:
            os.environ.pop("PRALINE_REF_BUDGET_MB", None)
        else:
            os.environ["PRALINE_REF_BUDGET_MB"] = old
        arena.close()


@pytest.mark.parametrize("setup", ["float-ref-tile", "float-ref-cell", "float-gaps", "wide-gaps"])
def test_dense_tiles_in_strip_ranges(nat, bba, monkeypatch, setup):
    """A task whose dense tile exceeds the launch budget is swept a range of strips per launch (plan_run_dense: the tile
    holds that range, the boundary column stays in scratch, a local alignment's running maximum travels through scores /
    end_cells).  With a 1 MiB budget every longer task of this list runs one or two strips per launch: scores-only and with
    paths, five modes, constant and per-position gap scores, rectangles - bit-identical to the same plans under the default
    budget, for each producer of the tiles (k_match_tile falls back to one thread per cell for such tasks; the fp32 MFMA
    chain; a 40-symbol alphabet without packed operands)."""
    rng = np.random.default_rng({"float-ref-tile": 1, "float-ref-cell": 2, "float-gaps": 3, "wide-gaps": 4}[setup])
    lens = [150, 33, 97, 64, 201, 31, 1, 130]
    if setup == "wide-gaps":
        A = 40
        S = rng.normal(0, 3, (A, A)).astype(np.float32)
        profs = []
        for L in lens:
            c = np.zeros((L, A), dtype=np.float32)
            for _ in range(4):
                c[np.arange(L), rng.integers(0, A, L)] += rng.integers(1, 4, L)
            profs.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    else:
        S = bba["S"]
        profs = [synth_profile(rng, L)[0] for L in lens]
    n = len(lens)
    pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
    rects = [[(3, 9, 30, 41), (60, 70, 20, 90)] if k % 3 == 0 else [] for k in range(len(pairs))]
    with_gaps = setup.endswith("gaps")
    gaps = [random_gaps(rng, L, exact=False) for L in lens]
    if setup.startswith("float-ref"):
        nat.set_match_mode("ref")
    if setup == "float-ref-cell":
        monkeypatch.setenv("PRALINE_NO_REFTILE", "1")
    want_producer = {"float-ref-tile": 1, "float-ref-cell": 2, "float-gaps": 3, "wide-gaps": 2}[setup]
    try:
        arena = nat.Arena(profs, S)
        if with_gaps:
            arena.set_gap_scores(gaps)

        def run(mode, want_paths, use_rects, ppg):
            plan = nat.Plan(arena, pairs, want_paths=want_paths, rects=rects if use_rects else None)
            assert plan.tile_producer() == want_producer
            if ppg:
                plan.run_gaps(mode)
            else:
                plan.run(mode, -9.5, -1.25)
            out = (plan.scores().copy(), [p.copy() for p in plan.paths()] if want_paths else None)
            plan.close()
            return out

        cases = [(mode, wp, False, ppg) for mode in MODES for wp in (False, True) for ppg in ((False, True) if with_gaps else (False,))]
        cases += [("local", True, True, ppg) for ppg in ((False, True) if with_gaps else (False,))]
        whole = {c: run(*c) for c in cases}
        monkeypatch.setenv("PRALINE_REFTILE_BUDGET_MB", "1")
        for c in cases:
            sc, paths = run(*c)
            assert np.array_equal(bits(sc), bits(whole[c][0])), (setup, c)
            if paths is not None:
                assert all(np.array_equal(x, y) for x, y in zip(paths, whole[c][1])), (setup, c)
        monkeypatch.delenv("PRALINE_REFTILE_BUDGET_MB")
        # scores-only and path plans agree, and a sample equals the oracle on the reference-order match scores
        for mode in MODES:
            for ppg in ((False, True) if with_gaps else (False,)):
                assert np.array_equal(bits(whole[(mode, False, False, ppg)][0]), bits(whole[(mode, True, False, ppg)][0])), (setup, mode, ppg)
        if setup != "float-gaps":
            for k in rng.choice(len(pairs), 10, replace=False):
                i, j = pairs[k]
                m = np.zeros((lens[i], lens[j]), dtype=np.float32)
                orc.cext_build_scores([profs[i]], [profs[j]], [orc.build_nonzero_matrix(profs[i])], [orc.build_nonzero_matrix(profs[j])], [S], m)
                for mode in MODES:
                    g1, g2 = orc.gap_arrays(lens[i], lens[j], (-9.5, -1.25))
                    s_or, p_or = orc.raw_pairwise_align(mode, m, g1, g2)
                    assert whole[(mode, True, False, False)][0][k] == np.float32(s_or), (setup, mode, i, j)
                    assert np.array_equal(whole[(mode, True, False, False)][1][k], p_or), (setup, mode, i, j)
                    if with_gaps:
                        s_or, p_or = orc.raw_pairwise_align(mode, m, gaps[i], gaps[j])
                        assert whole[(mode, True, False, True)][0][k] == np.float32(s_or), (setup, mode, i, j, "gaps")
                        assert np.array_equal(whole[(mode, True, False, True)][1][k], p_or), (setup, mode, i, j, "gaps")
        arena.close()
    finally:
        nat.set_match_mode(None)
