"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
libpraline_dp.so; the CPU oracle and the committed golden vectors are the checkers."""
import os

import numpy as np
import pytest

from conftest import MODES, load_golden, one_hot, synth_lengths, synth_profile
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GAPS = (-11.0, -1.0)


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    return native


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else a.dtype)


def oracle_dp_on_device_m(mode, p1, p2, S, gaps=GAPS, zero_idxs=None):
    """Oracle DP fed with the match scores in the fp32-MFMA (fma-chain) evaluation order."""
    m = orc.build_scores_fma([p1], [p2], [S])
    g1, g2 = orc.gap_arrays(m.shape[0], m.shape[1], gaps)
    return orc.raw_pairwise_align(mode, m, g1, g2, zero_idxs)


def oracle_dp_on_m(mode, m, gaps=GAPS, zero_idxs=None):
    """Oracle DP (fill, end cell, traceback) on a given match-score matrix."""
    g1, g2 = orc.gap_arrays(m.shape[0], m.shape[1], gaps)
    return orc.raw_pairwise_align(mode, np.ascontiguousarray(m), g1, g2, zero_idxs)


# ------------------------------------------------------------------------------------------------
# A. parity-layout entry points
# ------------------------------------------------------------------------------------------------
def test_build_scores_golden_and_order(nat, bba):
    d = load_golden("fill_small.npz")
    S = bba["S"]
    for n in range(int(d["n_cases"])):
        p = "c%03d_" % n
        p1, p2, m_ref = d[p + "p1"], d[p + "p2"], d[p + "m"]
        m = np.zeros_like(m_ref)
        nat.cext_build_scores([p1], [p2], None, None, [S], m)
        # bit-exact against the fma-order oracle; within 1e-5 relative of the reference order
        assert np.array_equal(bits(m), bits(orc.build_scores_fma([p1], [p2], [S]))), n
        if str(d[p + "kind"]).startswith("onehot") or str(d[p + "kind"]).startswith("mask"):
            assert np.array_equal(m, m_ref), n  # integer scoring: exact
        else:
            assert np.abs(m - m_ref).max() <= 1e-5 * max(1.0, np.abs(m_ref).max()), n


def test_build_scores_multiset_and_strides(nat, bba):
    d = load_golden("multitrack.npz")
    i, j = 0, 4
    for nsets, key in ((2, "s2_0_4_global_m"), (3, "s3_0_4_global_m")):
        i1 = [one_hot(bba["seqs"][i], 27), one_hot(bba["motif"][i], 2)]
        i2 = [one_hot(bba["seqs"][j], 27), one_hot(bba["motif"][j], 2)]
        ss = [bba["S"], bba["motif_matrix"]]
        if nsets == 3:
            i1.append(one_hot(bba["ss"][i], 3))
            i2.append(one_hot(bba["ss"][j], 3))
            ss.append(bba["ss_matrix"])
        # non-contiguous views: transposed storage and a column-strided output
        i1 = [np.asfortranarray(a) for a in i1]
        big = np.zeros((i1[0].shape[0], 2 * i2[0].shape[0]), dtype=np.float32)
        m = big[:, ::2]
        nat.cext_build_scores(i1, i2, None, None, ss, m)
        assert np.array_equal(m, d[key])
        assert not big[:, 1::2].any()


def test_align_fill_golden_bitexact(nat):
    d = load_golden("fill_small.npz")
    for n in range(int(d["n_cases"])):
        p = "c%03d_" % n
        mode = str(d[p + "mode"])
        o, t = d[p + "o0"].copy(), d[p + "t0"].copy()
        getattr(nat, "cext_align_" + mode)(d[p + "m"], d[p + "g1"], d[p + "g2"], o, t, d[p + "z"])
        assert np.array_equal(bits(o), bits(d[p + "o"])), (n, mode)
        assert np.array_equal(t, d[p + "t"]), (n, mode)
        zi = d[p + "z"] if d[p + "z"].any() else None
        score, path = nat.raw_align(mode, d[p + "m"], d[p + "g1"], d[p + "g2"], zi)
        assert score == float(d[p + "score"]), (n, mode)
        assert np.array_equal(path, d[p + "path"]), (n, mode)


def test_align_fill_random_vs_oracle(nat):
    rng = np.random.default_rng(11)
    for trial in range(10):
        L1, L2 = (int(v) for v in rng.integers(1, 200, 2))
        m = (rng.normal(size=(L1, L2)) * 5).astype(np.float32)
        if trial % 2:
            m = np.rint(m)
        g1 = rng.normal(-5, 2, (L1, 2)).astype(np.float32)
        g2 = rng.normal(-5, 2, (L2, 2)).astype(np.float32)
        zi = [(int(rng.integers(0, L1 + 1)), int(rng.integers(0, L2 + 1))) for _ in range(3 * trial)]
        for mode in MODES:
            o, t, z = orc.init_matrices(mode, g1, g2, zi)
            o2, t2 = o.copy(), t.copy()
            orc.CEXT_ALIGN_FUNCTIONS[mode](m, g1, g2, o, t, z)
            getattr(nat, "cext_align_" + mode)(m, g1, g2, o2, t2, z)
            assert np.array_equal(bits(o), bits(o2)) and np.array_equal(t, t2), (trial, mode)
            score, path = nat.raw_align(mode, m, g1, g2, z)
            s_ref, p_ref = orc.raw_pairwise_align(mode, m, g1, g2, zi)
            assert score == s_ref and np.array_equal(path, p_ref), (trial, mode)


# ------------------------------------------------------------------------------------------------
# B. batched path
# ------------------------------------------------------------------------------------------------
def all_pairs(n):
    return np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.int32)


@pytest.mark.parametrize("name,A,n", [("synthetic_c1.npz", 27, 8), ("synthetic_dna.npz", 15, 6)])
def test_batch_golden_scores_and_paths(nat, bba, name, A, n):
    d = load_golden(name)
    S = bba["S"] if A == 27 else d["matrix"]
    profs = [one_hot(d["seq%d" % i], A) for i in range(n)]
    arena = nat.Arena(profs, S)
    pairs = all_pairs(n)
    for mode in MODES:
        plan = nat.Plan(arena, pairs)
        plan.run(mode, *GAPS)
        assert np.array_equal(plan.scores().astype(np.float64), d["scores_" + mode]), mode
        plan.close()
        plan = nat.Plan(arena, pairs, want_paths=True)
        plan.run(mode, *GAPS)
        assert np.array_equal(plan.scores().astype(np.float64), d["scores_" + mode]), mode
        off = d["paths_off_" + mode]
        for k, path in enumerate(plan.paths()):
            assert np.array_equal(path, d["paths_" + mode][off[k]:off[k + 1]]), (mode, k)
        plan.close()
    arena.close()


def test_batch_kat_appendix_b(nat, bba):
    d = load_golden("kat_pairwise.npz")
    profs = [one_hot(s, 27) for s in bba["seqs"]]
    arena = nat.Arena(profs, bba["S"])
    pairs = all_pairs(5)
    for mode in MODES:
        plan = nat.Plan(arena, pairs, want_paths=True)
        plan.run(mode, *GAPS)
        sc, paths = plan.scores(), plan.paths()
        for k, (i, j) in enumerate(pairs):
            assert sc[k] == float(d["score_%d_%d_%s" % (i, j, mode)]), (mode, i, j)
            assert np.array_equal(paths[k], d["path_%d_%d_%s" % (i, j, mode)]), (mode, i, j)
        plan.close()
    arena.close()


def test_batch_float_profiles(nat, bba):
    """Float profile scoring: within 1e-5 relative of the reference (golden), and bit-identical to
    the oracle DP fed with the device-order match scores."""
    d = load_golden("profile_profile.npz")
    profs = [d["profile%d" % i] for i in range(5)]
    arena = nat.Arena(profs, bba["S"])
    pairs = all_pairs(5)
    for mode in MODES:
        plan = nat.Plan(arena, pairs, want_paths=True)
        kind = plan.match_kind()
        plan.run(mode, *GAPS)
        sc, paths = plan.scores(), plan.paths()
        for k, (i, j) in enumerate(pairs):
            ref = float(d["score_%d_%d_%s" % (i, j, mode)])
            assert abs(sc[k] - ref) <= 1e-5 * max(1.0, abs(ref)), (mode, i, j, sc[k], ref)
            # bit-identical to the oracle DP (fill, end cell, traceback) on the device's own match scores
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, kind))
            assert sc[k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(paths[k], p_or), (mode, i, j)
            # and the path equals the reference's on these inputs
            assert np.array_equal(paths[k], d["path_%d_%d_%s" % (i, j, mode)]), (mode, i, j)
        plan.close()
        # scores-only plan: match scores on the matrix pipe (f16 hi/lo split); the DP must be
        # bit-identical to the oracle DP on the device's own match scores, and the scores stay
        # within 1e-5 relative of the reference
        plan = nat.Plan(arena, pairs)
        kind = plan.match_kind()
        plan.run(mode, *GAPS)
        sc = plan.scores()
        plan.close()
        for k, (i, j) in enumerate(pairs):
            ref = float(d["score_%d_%d_%s" % (i, j, mode)])
            assert abs(sc[k] - ref) <= 1e-5 * max(1.0, abs(ref)), (mode, i, j, sc[k], ref)
            s_or, _ = oracle_dp_on_m(mode, arena.match_scores(i, j, kind))
            assert sc[k] == np.float32(s_or), (mode, i, j, kind)
    # the f16-split match scores against the reference's own evaluation order
    assert arena.info()["f16_terms"] in (2, 3)      # three terms: K-packed into four MFMAs (<= 21 active symbols) or six
    for (i, j) in ((0, 4), (2, 3)):
        m_ref = d["m_%d_%d" % (i, j)]
        for kind in (0, 1):
            m_dev = arena.match_scores(i, j, kind)
            assert np.abs(m_dev - m_ref).max() <= 2e-6 * np.abs(m_ref).max(), (i, j, kind)
    arena.close()


def test_exact_mode_for_integer_scoring(nat, bba):
    """One-hot x integer matrix: every operand is f16-representable, the matrix-pipe path runs its
    single-term exact mode and its match scores equal the fp32 chain's (and the reference's) bitwise."""
    profs = [one_hot(s, 27) for s in bba["seqs"]]
    arena = nat.Arena(profs, bba["S"])
    assert arena.info()["f16_terms"] == 1
    m0, m1 = arena.match_scores(0, 4, 0), arena.match_scores(0, 4, 1)
    assert np.array_equal(bits(m0), bits(m1))
    m_ref = np.zeros_like(m0)
    orc.cext_build_scores([profs[0]], [profs[4]], [orc.build_nonzero_matrix(profs[0])],
                          [orc.build_nonzero_matrix(profs[4])], [bba["S"]], m_ref)
    assert np.array_equal(m0, m_ref)
    arena.close()


@pytest.mark.parametrize("case", ["integer", "half_integer_scores", "half_integer_gaps", "odd_gaps", "dyadic_profiles"])
def test_traceback_tie_flags_exact_arithmetic(nat, bba, case, monkeypatch):
    """The single-term traceback instances take the tie flags from the predecessor states (valid when every DP
    value is exact in float32: one-hot rows, scores and gap scores on a common dyadic grid); anything else runs
    the instances that compare the candidate sums.  Scores and paths against the oracle in every mode (with
    Waterman-Eggert rectangles in local mode), and bitwise the same with the shortcut switched off."""
    rng = np.random.default_rng(17)
    N = 12
    lens = synth_lengths(rng, N, 70)
    lens[0], lens[1] = 1, 34
    S = np.array(bba["S"], dtype=np.float32)
    gaps = GAPS
    if case == "half_integer_scores":
        S = S * np.float32(0.5)
    elif case == "half_integer_gaps":
        gaps = (-10.5, -0.5)
    elif case == "odd_gaps":
        gaps = (-10.3, -0.7)          # not on a dyadic grid: the sums are rounded, the shortcut must not be taken
    if case == "dyadic_profiles":     # exact in f16 but not one-hot
        profs = []
        for L in lens:
            p = np.zeros((int(L), 27), dtype=np.float32)
            a, b = rng.integers(0, 20, int(L)), rng.integers(0, 20, int(L))
            p[np.arange(int(L)), a] += 0.5
            p[np.arange(int(L)), b] += 0.5
            profs.append(p)
    else:
        profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    arena = nat.Arena(profs, S)
    assert arena.info()["f16_terms"] == 1
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if (i + j) % 2 == 1], dtype=np.int32)
    rects = [[(3, 9, 2, 8)] if k % 3 == 0 else [] for k in range(len(pairs))]
    results = {}
    for shortcut in ("1", "0"):
        monkeypatch.setenv("PRALINE_NO_INTS", "0" if shortcut == "1" else "1")
        for mode in MODES:
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects if mode == "local" else None)
            pk = plan.match_kind()
            plan.run(mode, *gaps)
            results[(shortcut, mode)] = (plan.scores().copy(), [p.copy() for p in plan.paths()])
            plan.close()
            if shortcut == "0":
                continue
            sc, paths = results[(shortcut, mode)]
            for k in range(len(pairs)):
                i, j = pairs[k]
                zero = None
                if mode == "local" and rects[k]:
                    zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)
                            if y <= lens[i] and x <= lens[j]]
                s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, pk), gaps, zero)
                assert sc[k] == np.float32(s_or), (case, mode, i, j)
                assert np.array_equal(paths[k], p_or), (case, mode, i, j)
    for mode in MODES:
        a, b = results[("1", mode)], results[("0", mode)]
        assert np.array_equal(bits(a[0]), bits(b[0])), (case, mode)
        assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1])), (case, mode)
    arena.close()


@pytest.mark.parametrize("kind", ["onehot", "profile"])
def test_path_plan_execution_forms_agree(nat, bba, kind, monkeypatch):
    """A path plan runs in chain mode (one wave per task and strip; small plans), in task mode, or in task mode cut
    into chunks that fit a traceback-plane budget.  Same scores, end cells and paths in all three, every mode,
    with rectangles in local mode; spot-checked against the oracle."""
    rng = np.random.default_rng(23)
    N = 26
    lens = synth_lengths(rng, N, 110)
    lens[0], lens[1], lens[2] = 1, 32, 65
    profs = [one_hot(rng.integers(0, 20, int(L)), 27) if kind == "onehot" else synth_profile(rng, int(L))[0] for L in lens]
    arena = nat.Arena(profs, bba["S"])
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
    rects = [[(2, 20, 3, 30)] if k % 4 == 0 else ([(5, 9, 1, 4), (30, 60, 40, 64)] if k % 4 == 1 else [])
             for k in range(len(pairs))]
    # (plain sequences: the strip kernels with PRALINE_TB_QUAD=0; "quad" = k_dp_quad_tb, their default, whole and in chunks)
    # (... and k_dp_pk16_tb, the default of integer scoring: chain mode, one wave per task, chunks)
    forms = {"chain": {"PRALINE_TB_QUAD": "0"}, "tasks": {"PRALINE_TB_QUAD": "0", "PRALINE_NO_CHAIN": "1"},
             "chunks": {"PRALINE_TB_QUAD": "0", "PRALINE_NO_CHAIN": "1", "PRALINE_TB_BUDGET_MB": "1"},
             "quad": {"PRALINE_TB_QUAD": "1", "PRALINE_TB_PK16": "0"}, "quad chunks": {"PRALINE_TB_QUAD": "1", "PRALINE_TB_PK16": "0", "PRALINE_TB_BUDGET_MB": "1"},
             "pk16 chain": {}, "pk16 tasks": {"PRALINE_NO_CHAIN": "1"}, "pk16 chunks": {"PRALINE_NO_CHAIN": "1", "PRALINE_TB_BUDGET_MB": "1"},
             "pk16 chain chunks": {"PRALINE_TB_BUDGET_MB": "1"}}
    out = {}
    for form, env in forms.items():
        for key in ("PRALINE_NO_CHAIN", "PRALINE_TB_BUDGET_MB", "PRALINE_TB_QUAD", "PRALINE_TB_PK16"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        for mode in MODES:
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects if mode == "local" else None)
            pk = plan.match_kind()
            plan.run(mode, *GAPS)
            out[(form, mode)] = (plan.scores().copy(), [p.copy() for p in plan.paths()])
            if kind == "onehot":
                want = "k_dp_quad_tb" if form.startswith("quad") else ("k_dp_pk16_tb" if form.startswith("pk16") else "k_dp_split16_tb")
                assert plan.kernel_name().startswith(want), (form, plan.kernel_name())
                if form.startswith("pk16"):
                    assert plan.kernel_name().endswith(", true>") == ("chain" in form), (form, plan.kernel_name())
            plan.close()
    for key in ("PRALINE_NO_CHAIN", "PRALINE_TB_BUDGET_MB", "PRALINE_TB_QUAD", "PRALINE_TB_PK16"):
        monkeypatch.delenv(key, raising=False)
    for mode in MODES:
        ref_sc, ref_paths = out[("chain", mode)]
        for form in ("tasks", "chunks", "quad", "quad chunks", "pk16 chain", "pk16 tasks", "pk16 chunks", "pk16 chain chunks"):
            sc, paths = out[(form, mode)]
            assert np.array_equal(bits(sc), bits(ref_sc)), (form, mode)
            assert all(np.array_equal(a, b) for a, b in zip(paths, ref_paths)), (form, mode)
        for k in range(0, len(pairs), 37):
            i, j = pairs[k]
            zero = None
            if mode == "local" and rects[k]:
                zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)
                        if y <= lens[i] and x <= lens[j]]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, pk), GAPS, zero)
            assert ref_sc[k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(ref_paths[k], p_or), (mode, i, j)
    arena.close()


def test_pipeline_two_pass_paths_equal_single_pass(nat, bba, monkeypatch):
    """Global alignments with paths of float-profile plans run two passes with the PIPELINE kernel as the forward fill
    (k_dp_pipe<..., KEEP>: kept strip columns, (M, U, L) of every PRALINE_KEEP_BH-th row, corner states) and
    k_trace_recompute on the blocks each path crosses.  Scores, end states and paths must equal the single pass (chain mode,
    PRALINE_TB_PIPE=0) bit for bit and the oracle's walk on the device's match scores: ragged lengths (1 .. 3 blocks of
    rows, sequences shorter than a strip), sets with empty lanes, a pair list with holes, several items per set."""
    rng = np.random.default_rng(41)
    N = 70
    lens = synth_lengths(rng, N, 90)
    lens[:6] = [1, 2, 31, 33, 35, 37]
    lens[6], lens[7] = 150, 73
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    arena = nat.Arena(profs, bba["S"])
    allp = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
    # (a few holes only: lists that leave the sets of 32 sequences one much emptier than the per-column tasks would be keep
    # the task schedule - sched.cpp, build_pipe_schedule)
    for case, pairs in (("all ordered pairs", allp), ("with holes", allp[rng.random(len(allp)) < 0.97])):
        out = {}
        for form, env in (("single", {"PRALINE_TB_PIPE": "0"}), ("pipeline", {"PRALINE_TB_PIPE": "1", "PRALINE_PIPE_MIN_TASKS": "1"})):
            for key, val in env.items():
                monkeypatch.setenv(key, val)
            plan = nat.Plan(arena, pairs, want_paths=True)
            pk = plan.match_kind()
            plan.run("global", *GAPS)
            out[form] = (plan.scores().copy(), [p.copy() for p in plan.paths()], plan.kernel_name())
            # a second run of the same plan (the analytic column is reused) and another gap pair (it is rewritten)
            plan.run("global", -7.5, -0.5)
            out[form + "2"] = (plan.scores().copy(), [p.copy() for p in plan.paths()])
            plan.close()
        assert out["pipeline"][2].startswith("k_dp_pipe<") and "true>" in out["pipeline"][2], out["pipeline"][2]
        assert not out["single"][2].startswith("k_dp_pipe"), out["single"][2]
        for a, b in (("single", "pipeline"), ("single2", "pipeline2")):
            assert np.array_equal(bits(out[a][0]), bits(out[b][0])), (case, a)
            assert all(np.array_equal(x, y) for x, y in zip(out[a][1], out[b][1])), (case, a)
        for k in range(0, len(pairs), 97):
            i, j = pairs[k]
            s_or, p_or = oracle_dp_on_m("global", arena.match_scores(i, j, pk), GAPS, None)
            assert out["pipeline"][0][k] == np.float32(s_or), (case, i, j)
            assert np.array_equal(out["pipeline"][1][k], p_or), (case, i, j)
    arena.close()


def test_quad_layout_paths_equal_the_strip_kernels(nat, bba, monkeypatch):
    """Alignments with paths of plain sequences run k_dp_quad_tb (dp_quad.hip.h: 16 pairs per wave, four lanes of 8 columns
    per pair, two rows per step, its own traceback planes).  Scores, end cells and paths must equal the strip kernels'
    (PRALINE_TB_QUAD=0) bit for bit and the oracle's: five modes; zero rectangles held in registers (<= 4 per pair) and as
    per-row mask words (more: on the strip layout those plans run the dense-tile instances); gap scores on and off the integer grid
    (tie flags from the predecessor states / from the candidate sums); lengths around the 8-column quarters, the 32-column
    strips and the two-row steps; sequences of one residue; lanes without a pair."""
    rng = np.random.default_rng(97)
    lens = np.array([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 129, 47, 250, 71, 5, 24, 25, 96, 97, 40, 41, 200, 13, 58, 77, 12, 35, 36])
    profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    N = len(lens)
    arena = nat.Arena(profs, bba["S"])
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < 0.6]

    def rect_lists(nmax):
        out = []
        for (i, j) in pairs:
            rl = []
            for _ in range(int(rng.integers(0, nmax + 1))):
                y0, x0 = int(rng.integers(1, lens[i] + 1)), int(rng.integers(1, lens[j] + 1))
                rl.append((y0, min(int(lens[i]), y0 + int(rng.integers(0, 12))), x0, min(int(lens[j]), x0 + int(rng.integers(0, 30)))))
            out.append(rl)
        return out
    cases = [(mode, gaps, None) for mode in MODES for gaps in (GAPS, (-10.3, -1.7))]
    cases += [("local", GAPS, rect_lists(3)), ("local", (-7.5, -0.5), rect_lists(9)), ("global", GAPS, rect_lists(6))]
    monkeypatch.setenv("PRALINE_TB_PK16", "0")     # (integer scoring would take k_dp_pk16_tb)
    for mode, gaps, rects in cases:
        res = {}
        for quad in ("0", "1"):
            monkeypatch.setenv("PRALINE_TB_QUAD", quad)
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
            pk = plan.match_kind()
            plan.run(mode, *gaps)
            res[quad] = (plan.scores().copy(), [p.copy() for p in plan.paths()], plan.kernel_name())
            plan.close()
        assert res["1"][2].startswith("k_dp_quad_tb") and not res["0"][2].startswith("k_dp_quad_tb"), (res["0"][2], res["1"][2])
        assert np.array_equal(bits(res["0"][0]), bits(res["1"][0])), (mode, gaps, rects is not None)
        bad = [k for k in range(len(pairs)) if not np.array_equal(res["0"][1][k], res["1"][1][k])]
        assert not bad, (mode, gaps, rects is not None, pairs[bad[:3]].tolist())
        for k in range(0, len(pairs), 41):
            i, j = pairs[k]
            zero = None
            if rects is not None and rects[k]:
                zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(int(i), int(j), pk), gaps, zero)
            assert res["1"][0][k] == np.float32(s_or), (mode, gaps, i, j)
            assert np.array_equal(res["1"][1][k], p_or), (mode, gaps, i, j)
    arena.close()


def test_packed_int16_paths_equal_the_float_kernels(nat, bba, monkeypatch):
    """Integer scoring within int16 runs k_dp_pk16_tb (dp_pk16.hip.h: two pairs per lane, v_pk_*_i16 with saturation, sign bytes
    gathered with v_perm_b32, traceback layout 3).  Scores, end cells and paths must equal k_dp_quad_tb's (PRALINE_TB_PK16=0) bit
    for bit and the oracle's: five modes, integer / half-integer gap scores (scale 1 / 2) and open == extend, rectangles in
    registers, lengths around the quarters / strips / two-row steps, lanes and half lanes without a pair, sequence 0 in the
    upper half of a lane.  Gap scores off the grid (-10.3, -1.7) and scores beyond int16 fall back to the float kernels on the
    same plan."""
    rng = np.random.default_rng(131)
    lens = np.array([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 129, 47, 250, 71, 5, 24, 25, 96, 97, 40, 41, 200, 13, 58, 77, 12, 35, 36,
                     300, 18, 19, 20, 21, 22, 23, 110, 111, 112, 45, 46, 66, 67, 68, 69])
    profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    N = len(lens)
    arena = nat.Arena(profs, bba["S"])
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < 0.7]

    def rect_lists(nmax):
        out = []
        for (i, j) in pairs:
            rl = []
            for _ in range(int(rng.integers(0, nmax + 1))):
                y0, x0 = int(rng.integers(1, lens[i] + 1)), int(rng.integers(1, lens[j] + 1))
                rl.append((y0, min(int(lens[i]), y0 + int(rng.integers(0, 12))), x0, min(int(lens[j]), x0 + int(rng.integers(0, 30)))))
            out.append(rl)
        return out
    cases = [(mode, gaps, None) for mode in MODES for gaps in (GAPS, (-7.5, -0.5), (-4.0, -4.0))]
    cases += [("local", GAPS, rect_lists(3)), ("local", (-7.5, -0.5), rect_lists(4)), ("global", GAPS, rect_lists(4)), ("semiglobal_both", GAPS, rect_lists(2))]
    cases += [("global", (-10.3, -1.7), None), ("local", (-60.0, -1.0), None)]     # off the grid; (L1 + L2 + 2) * 60 > 32 000
    for mode, gaps, rects in cases:
        res = {}
        monkeypatch.setenv("PRALINE_TB_QUAD", "1")      # (the comparison: k_dp_quad_tb also for a plan of this size)
        for pk16 in ("0", "1"):
            monkeypatch.setenv("PRALINE_TB_PK16", pk16)
            monkeypatch.setenv("PRALINE_NO_CHAIN", "1" if mode in ("global", "semiglobal_two") and rects is None else "0")   # one wave per task / chain mode
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
            pk = plan.match_kind()
            plan.run(mode, *gaps)
            res[pk16] = (plan.scores().copy(), [p.copy() for p in plan.paths()], plan.kernel_name())
            plan.close()
        fits = gaps not in ((-10.3, -1.7), (-60.0, -1.0))
        assert res["1"][2].startswith("k_dp_pk16_tb") == fits and res["0"][2].startswith("k_dp_quad_tb"), (gaps, res["0"][2], res["1"][2])
        assert not fits or res["1"][2].endswith(", true>") == (not (mode in ("global", "semiglobal_two") and rects is None)), res["1"][2]
        assert np.array_equal(bits(res["0"][0]), bits(res["1"][0])), (mode, gaps, rects is not None)
        bad = [k for k in range(len(pairs)) if not np.array_equal(res["0"][1][k], res["1"][1][k])]
        assert not bad, (mode, gaps, rects is not None, pairs[bad[:3]].tolist())
        for k in range(0, len(pairs), 97):
            i, j = pairs[k]
            zero = None
            if rects is not None and rects[k]:
                zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(int(i), int(j), pk), gaps, zero)
            assert res["1"][0][k] == np.float32(s_or), (mode, gaps, i, j)
            assert np.array_equal(res["1"][1][k], p_or), (mode, gaps, i, j)
    arena.close()
    for key in ("PRALINE_TB_PK16", "PRALINE_TB_QUAD", "PRALINE_NO_CHAIN"):
        monkeypatch.delenv(key)


def test_batch_waterman_eggert_masks(nat, bba):
    """LocalMasterSlaveAligner's inner calls (praline/component/preprofile.py:227-267)."""
    d = load_golden("preprofile.npz")
    profs = [one_hot(s, 27) for s in bba["seqs"]]
    arena = nat.Arena(profs, bba["S"])
    for key, master, iters in (("local_m0_", 0, 2), ("local_m4_", 4, 2), ("local_we3_m0_", 0, 3)):
        slaves = [k for k in range(5) if k != master]
        pairs = np.array([(master, s) for s in slaves], dtype=np.int32)
        rects = [[] for _ in slaves]
        for it in range(iters):
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
            plan.run("local", *GAPS)
            sc, paths = plan.scores(), plan.paths()
            plan.close()
            for q in range(len(slaves)):
                c = q * iters + it
                assert sc[q] == float(d[key + "call%d_score" % c]), (key, c)
                assert np.array_equal(paths[q], d[key + "call%d_path" % c]), (key, c)
                p = paths[q]
                rects[q] = rects[q] + [(int(p[:, 0].min()), int(p[:, 0].max()),
                                        int(p[:, 1].min()), int(p[:, 1].max()))]
    arena.close()


@pytest.mark.parametrize("kind", ["onehot", "profile"])
def test_batch_random_vs_oracle(nat, bba, kind):
    rng = np.random.default_rng(3)
    N = 40
    lens = synth_lengths(rng, N, 90)
    lens[0], lens[1], lens[2] = 1, 32, 33  # edge lengths: single residue, exact strip, strip + 1
    if kind == "onehot":
        profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    else:
        profs = [synth_profile(rng, int(L))[0] for L in lens]
    S = bba["S"]
    arena = nat.Arena(profs, S)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j and (i * 7 + j) % 5 == 0],
                     dtype=np.int32)
    for mode in MODES:
        plan = nat.Plan(arena, pairs, want_paths=True)
        pk = plan.match_kind()
        plan.run(mode, *GAPS)
        sc, paths = plan.scores(), plan.paths()
        plan.close()
        plan0 = nat.Plan(arena, pairs)
        mk = plan0.match_kind()      # how the scores-only kernel evaluates m (honours PRALINE_MM / PRALINE_KERNEL)
        plan0.run(mode, *GAPS)
        sc0 = plan0.scores()
        plan0.close()
        if kind == "onehot":
            assert np.array_equal(bits(sc), bits(sc0)), mode  # integer scoring: all kernels agree bitwise
        else:
            assert np.abs(sc - sc0).max() <= 1e-5 * np.abs(sc).max(), mode
        for k in range(0, len(pairs), 3):
            i, j = pairs[k]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, pk))
            assert sc[k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(paths[k], p_or), (mode, i, j)
            s16, _ = oracle_dp_on_m(mode, arena.match_scores(i, j, mk))
            assert sc0[k] == np.float32(s16), (mode, i, j)
    arena.close()


@pytest.mark.parametrize("gaps", [(0.0, 0.0), (-3.0, 0.0), (0.0, -2.0), (-1.0, -1.0), (-4.5, -0.25)])
def test_unusual_gap_scores_vs_oracle(nat, bba, gaps):
    """Zero gap scores make U and L tie with M everywhere (end-cell and traceback tie-breaks; the local
    kernel's M-only maximum is valid for strictly negative gap scores only and must not be used here)."""
    rng = np.random.default_rng(int(-gaps[0] * 8 - gaps[1] * 64) + 5)
    N = 14
    lens = synth_lengths(rng, N, 60)
    lens[0], lens[1] = 1, 33
    profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    arena = nat.Arena(profs, bba["S"])
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if (i + 2 * j) % 3 == 0], dtype=np.int32)
    for mode in MODES:
        plan = nat.Plan(arena, pairs, want_paths=True)
        pk = plan.match_kind()
        plan.run(mode, *gaps)
        sc, paths = plan.scores(), plan.paths()
        plan.close()
        plan0 = nat.Plan(arena, pairs)
        plan0.run(mode, *gaps)
        assert np.array_equal(bits(sc), bits(plan0.scores())), (mode, gaps)
        plan0.close()
        for k in range(len(pairs)):
            i, j = pairs[k]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, pk), gaps)
            assert sc[k] == np.float32(s_or), (mode, gaps, i, j)
            assert np.array_equal(paths[k], p_or), (mode, gaps, i, j)
    arena.close()


@pytest.mark.parametrize("kind", ["onehot", "profile", "dna"])
def test_operand_stream_variants_agree_bitwise(nat, bba, kind, monkeypatch):
    """k_dp_split16 has three sources for the sequence-one operands (LDS one-hot table, LDS-staged
    DMA stream, per-lane global loads): same arithmetic, so the scores must agree bit for bit."""
    rng = np.random.default_rng(11)
    if kind == "dna":
        N, mu, A = 24, 700, 15
        S = load_golden("synthetic_dna.npz")["matrix"]
        lens = synth_lengths(rng, N, mu)
        profs = [one_hot(rng.integers(0, 4, L), A) for L in lens]
    else:
        N, mu, A = 70, 150, 27
        S = bba["S"]
        lens = synth_lengths(rng, N, mu)
        lens[0], lens[1], lens[2], lens[3] = 1, 5, 32, 33
        if kind == "onehot":
            profs = [one_hot(rng.integers(0, 20, L), A) for L in lens]
        else:
            profs = [synth_profile(rng, int(L))[0] for L in lens]
    pairs = all_pairs(N)
    results = {}
    # default: one-hot arenas look their match scores up (no MFMA; shared waves here, four singles with PRALINE_NO_W2);
    # "table": the one-hot operand table feeding MFMAs; "staged" / "lanes": the operand streams of float profiles
    # (batches of this size would run the flag-free chain fill: held off for the score kernels' variants, and compared
    # with them as the variant "chain")
    for variant, env in (("default", {}), ("lookup_singles", {"PRALINE_NO_W2": "1"}), ("table", {"PRALINE_NO_LOOKUP": "1"}),
                         ("table_singles", {"PRALINE_NO_LOOKUP": "1", "PRALINE_NO_W2": "1"}), ("staged", {"PRALINE_NO_ONEHOT": "1"}),
                         ("lanes", {"PRALINE_NO_ONEHOT": "1", "PRALINE_NO_STAGE": "1"}), ("chain", {"PRALINE_SCORES_CHAIN": "1"})):
        for k in ("PRALINE_NO_ONEHOT", "PRALINE_NO_STAGE", "PRALINE_NO_LOOKUP", "PRALINE_NO_W2"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("PRALINE_SCORES_CHAIN", "0")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        arena = nat.Arena(profs, S)
        plan = nat.Plan(arena, pairs)
        for mode in MODES:
            plan.run(mode, *GAPS)
            results[(variant, mode)] = plan.scores().copy()
        if kind != "profile" and variant in ("default", "lookup_singles") and not any(os.environ.get(k) for k in ("PRALINE_KERNEL", "PRALINE_MM")):
            assert ", 3, " in plan.kernel_name(), plan.kernel_name()       # the lookup instances ran
        if variant == "chain" and not any(os.environ.get(k) for k in ("PRALINE_KERNEL", "PRALINE_MM")):
            assert "true, true, 0>" in plan.kernel_name(), plan.kernel_name()   # k_dp_split16_tb<..., CHAIN, TWOPASS>: no flags
        plan.close()
        arena.close()
    for mode in MODES:
        ref = results[("lanes", mode)]
        assert np.isfinite(ref).all()
        for variant in ("default", "lookup_singles", "table", "table_singles", "staged", "chain"):
            assert np.array_equal(bits(results[(variant, mode)]), bits(ref)), (kind, variant, mode)


@pytest.mark.parametrize("chain", ["0", "1"])
@pytest.mark.parametrize("kind", ["onehot", "profile"])
def test_shared_wave_boundaries_vs_oracle(nat, bba, kind, chain, monkeypatch):
    """Small batches run on four-wave workgroups whose waves share tasks (1, 2 or 4 waves per task,
    chosen from the number of 12-row iterations and of 32-column strips) - chain "0" - or, where the schedule's estimate
    favours it, on the flag-free chain fill with one wave per task and strip - chain "1".  Lengths sit on both sides
    of every threshold; every score is checked against the oracle DP on the device's match scores
    (integer scoring: against the reference order too)."""
    monkeypatch.setenv("PRALINE_SCORES_CHAIN", chain)
    rng = np.random.default_rng(17)
    S = bba["S"]
    # sequence two: 1..5 strips (+- 1 column around the strip edges); sequence one: 2..9 iterations
    lens_two = [31, 32, 33, 63, 64, 65, 96, 97, 127, 128, 129, 160]
    lens_one = [23, 24, 25, 36, 37, 47, 48, 49, 60, 61, 84, 85, 96, 97, 108]
    lens = lens_one + lens_two
    if kind == "onehot":
        profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    else:
        profs = [synth_profile(rng, int(L))[0] for L in lens]
    n1 = len(lens_one)
    pairs = np.array([(i, n1 + j) for j in range(len(lens_two)) for i in range(n1)], dtype=np.int32)
    arena = nat.Arena(profs, S)
    plan = nat.Plan(arena, pairs)
    mk = plan.match_kind()
    for mode in MODES:
        plan.run(mode, *GAPS)
        sc = plan.scores()
        for k in range(len(pairs)):
            i, j = pairs[k]
            s_or, _ = oracle_dp_on_m(mode, arena.match_scores(i, j, mk))
            assert sc[k] == np.float32(s_or), (kind, mode, lens[i], lens[j])
            if kind == "onehot" and k % 7 == 0:
                s_ref = orc.pairwise_score_fast(mode, profs[i], profs[j], S, *GAPS)
                assert sc[k] == np.float32(s_ref), (mode, lens[i], lens[j])
    plan.close()
    arena.close()


def test_wide_concatenated_alphabet(nat, bba):
    """Two track sets whose concatenated alphabet (27 + 12 = 39 symbols) exceeds 32 while the ACTIVE symbols
    (20 residues + 10 states) do not: parity entry point and batched plan against the multi-set oracle
    (cext.c:389-420 sums the sets)."""
    rng = np.random.default_rng(31)
    S1 = bba["S"]
    S2 = np.zeros((12, 12), dtype=np.float32)
    S2[:10, :10] = rng.integers(-3, 6, (10, 10)).astype(np.float32)
    S2 = np.maximum(S2, S2.T)
    L1, L2 = 57, 70
    a1, a2 = rng.integers(0, 20, L1), rng.integers(0, 20, L2)
    b1, b2 = rng.integers(0, 10, L1), rng.integers(0, 10, L2)
    i1s, i2s = [one_hot(a1, 27), one_hot(b1, 12)], [one_hot(a2, 27), one_hot(b2, 12)]
    m = np.zeros((L1, L2), dtype=np.float32)
    nat.cext_build_scores(i1s, i2s, None, None, [S1, S2], m)
    want = S1[np.ix_(a1, a2)] + S2[np.ix_(b1, b2)]
    assert np.array_equal(m, want)
    cat = lambda x, y: np.concatenate([x, y], axis=1)
    Sbig = np.zeros((39, 39), dtype=np.float32)
    Sbig[:27, :27] = S1
    Sbig[27:, 27:] = S2
    arena = nat.Arena([cat(*i1s), cat(*i2s)], Sbig)
    assert arena.info()["n_active"] == 30
    for mode in MODES:
        plan = nat.Plan(arena, np.array([(0, 1)], dtype=np.int32), want_paths=True)
        plan.run(mode, *GAPS)
        s_or, p_or = oracle_dp_on_m(mode, want)
        assert plan.scores()[0] == np.float32(s_or), mode
        assert np.array_equal(plan.paths()[0], p_or), mode
        plan.close()
    arena.close()


def test_batch_c2_slice_properties(nat, bba):
    """BASELINE config 1 shape (256 x ~400 aa profiles, all pairs, global): properties that do not
    need the oracle at full size + oracle spot checks."""
    rng = np.random.default_rng(2)
    N = 256
    lens = synth_lengths(rng, N, 400)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    S = bba["S"]
    arena = nat.Arena(profs, S)
    pairs = all_pairs(N)
    plan = nat.Plan(arena, pairs)
    mk = plan.match_kind()
    plan.run("global", *GAPS)
    sc = plan.scores()
    plan.close()
    assert np.isfinite(sc).all()
    # symmetric matrix + equal gap models: score(i, j) ~ score(j, i); the two orientations round
    # the float match scores differently (P_i . Q_j vs P_j . Q_i), so 1e-5 relative, not bitwise
    sub = pairs[::97]
    plan = nat.Plan(arena, sub[:, ::-1].copy())
    plan.run("global", *GAPS)
    rev = plan.scores()
    assert np.abs(rev - sc[::97]).max() <= 1e-5 * np.abs(sc[::97]).max()
    plan.close()
    # self alignment of a one-hot sequence scores the sum of its diagonal entries (no gaps)
    # local >= semiglobal_both >= global for every pair
    plan = nat.Plan(arena, sub)
    plan.run("local", *GAPS)
    loc = plan.scores()
    plan.run("semiglobal_both", *GAPS)
    semi = plan.scores()
    plan.close()
    # semiglobal frees end gaps of the global alignment; local is NOT an upper bound of semiglobal in the
    # reference (penalised boundary in local mode, align.py:371-385), only of 0
    assert (semi >= sc[::97]).all() and (loc >= 0).all()
    for k in range(0, len(pairs), 4001):
        i, j = pairs[k]
        s_or, _ = oracle_dp_on_m("global", arena.match_scores(i, j, mk))
        assert sc[k] == np.float32(s_or)
        s_ref = orc.pairwise_score_fast("global", profs[i], profs[j], S, *GAPS)
        assert abs(sc[k] - s_ref) <= 1e-5 * abs(s_ref)   # vs the reference evaluation order
    arena.close()


def test_edge_cases_vs_oracle(nat, bba):
    """Empty pair list, self and duplicate pairs, length-1 sequences, a hopeless local alignment (nothing
    scores above 0) and one long pair (3 000 x 9 000, 283 strips, shared by four waves): scores and paths
    against the oracle in all modes."""
    S = bba["S"]
    rng = np.random.default_rng(41)
    arena = nat.Arena([one_hot([3], 27), one_hot([3, 4], 27)], S)
    plan = nat.Plan(arena, np.zeros((0, 2), np.int32))
    plan.run("global", *GAPS)
    assert plan.scores().shape == (0,)
    plan.close()
    arena.close()
    # W (index of tryptophan) against cysteine-free junk: BLOSUM62 off-diagonals here are all negative
    vals = [np.array([3]), np.array([7]), rng.integers(0, 20, 40), np.full(25, 17), np.full(31, 4), rng.integers(0, 20, 40)]
    profs = [one_hot(v, 27) for v in vals]
    arena = nat.Arena(profs, S)
    pairs = np.array([(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (2, 2), (2, 5), (2, 5), (3, 4), (4, 3), (0, 3)], dtype=np.int32)
    for mode in MODES:
        plan = nat.Plan(arena, pairs, want_paths=True)
        plan.run(mode, *GAPS)
        sc, paths = plan.scores(), plan.paths()
        plan.close()
        plan = nat.Plan(arena, pairs)
        plan.run(mode, *GAPS)
        sc0 = plan.scores()
        plan.close()
        assert np.array_equal(bits(sc), bits(sc0)), mode
        for k, (i, j) in enumerate(pairs):
            s_or, p_or = orc.pairwise_score_fast(mode, profs[i], profs[j], S, *GAPS, want_path=True)
            assert sc[k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(paths[k], p_or), (mode, i, j)
    arena.close()
    # one long pair: 283 strips x 3000 rows; a single task, pipelined by four waves of one workgroup
    d = load_golden("synthetic_dna.npz")
    Sd = d["matrix"]
    a, b = rng.integers(0, 4, 3000), rng.integers(0, 4, 9040)
    b[100:2900] = a[60:2860]                 # a long common stretch so that local / semiglobal paths are long
    pa, pb = one_hot(a, 15), one_hot(b, 15)
    arena = nat.Arena([pa, pb], Sd)
    for mode in MODES:
        plan = nat.Plan(arena, np.array([(0, 1)], np.int32))
        plan.run(mode, *GAPS)
        s_dev = plan.scores()[0]
        plan.close()
        plan = nat.Plan(arena, np.array([(0, 1)], np.int32), want_paths=True)
        plan.run(mode, *GAPS)
        s_tb, p_tb = plan.scores()[0], plan.paths()[0]
        plan.close()
        s_or, p_or = orc.pairwise_score_fast(mode, pa, pb, Sd, *GAPS, want_path=True)
        assert s_dev == np.float32(s_or) and s_tb == np.float32(s_or), mode
        assert np.array_equal(p_tb, p_or), mode
    arena.close()


@pytest.mark.parametrize("kind", ["onehot", "profile"])
def test_premultiply_relaunch_is_idempotent(nat, bba, kind):
    """praline_arena_premultiply (pack + P.S^T + f16 split in one launch) must reproduce exactly what arena
    creation computed with the separate kernels: same scores, same match-score matrices."""
    rng = np.random.default_rng(53)
    lens = [1, 31, 32, 33, 64, 97, 130]
    profs = [one_hot(rng.integers(0, 20, L), 27) if kind == "onehot" else synth_profile(rng, L)[0] for L in lens]
    arena = nat.Arena(profs, bba["S"])
    pairs = all_pairs(len(lens))
    plan = nat.Plan(arena, pairs)
    plan.run("global", *GAPS)
    before = plan.scores().copy()
    m_before = [arena.match_scores(2, 6, k).copy() for k in (0, 1)]
    for _ in range(2):
        arena.premultiply()
    plan.run("global", *GAPS)
    assert np.array_equal(bits(plan.scores()), bits(before))
    for k in (0, 1):
        assert np.array_equal(bits(arena.match_scores(2, 6, k)), bits(m_before[k]))
    plan.close()
    arena.close()


def test_two_pass_paths_equal_single_pass(nat, bba, monkeypatch):
    """The two-pass scheme for alignments with paths (flag-free forward fill with kept strip boundaries and row
    checkpoints + k_trace_recompute, dp_trace2.hip.h) against the single pass and the oracle: identical scores and
    paths in every mode, with Waterman-Eggert rectangles, one-hot (one-hot table operands) and float profiles, lengths
    that straddle the 32-row blocks and the 32-column strips."""
    monkeypatch.setenv("PRALINE_TB_QUAD", "0")   # (plain sequences would take k_dp_quad_tb: this test is about the strip kernels' two passes)
    rng = np.random.default_rng(41)
    lens = [1, 31, 32, 33, 64, 65, 97, 130, 200, 47]
    sets = {"onehot": [one_hot(rng.integers(0, 20, L), 27) for L in lens],
            "float": [synth_profile(rng, L)[0] for L in lens]}
    pairs = np.array([(i, j) for i in range(len(lens)) for j in range(len(lens)) if i != j], dtype=np.int32)
    rects = [[(3, 40, 2, 37)] if k % 3 == 0 else ([(50, 70, 20, 90), (1, 5, 60, 64)] if k % 3 == 1 else []) for k in range(len(pairs))]
    for kind, profs in sets.items():
        arena = nat.Arena(profs, bba["S"])
        for mode in MODES:
            for use_rects in ((False, True) if mode == "local" else (False,)):
                res = {}
                for two in ("0", "2"):
                    monkeypatch.setenv("PRALINE_TB_TWOPASS", two)
                    plan = nat.Plan(arena, pairs, want_paths=True, rects=rects if use_rects else None)
                    mk = plan.match_kind()
                    plan.run(mode, *GAPS)
                    res[two] = (plan.scores().copy(), [p.copy() for p in plan.paths()], plan.kernel_name())
                    plan.close()
                if not res["2"][2].endswith("true>") and os.environ.get("PRALINE_MM"):
                    pytest.skip("the two-pass kernels need the f16 operand layouts (switched off by the environment)")
                assert res["0"][2] != res["2"][2] and res["2"][2].endswith("true>"), res["2"][2]
                assert np.array_equal(bits(res["0"][0]), bits(res["2"][0])), (kind, mode, use_rects)
                for k, (i, j) in enumerate(pairs):
                    assert np.array_equal(res["0"][1][k], res["2"][1][k]), (kind, mode, use_rects, i, j)
                for k in range(0, len(pairs), 7):
                    i, j = pairs[k]
                    zero = [(y, x) for (y0, y1, x0, x1) in (rects[k] if use_rects else []) for y in range(y0, min(y1, lens[i]) + 1)
                            for x in range(x0, min(x1, lens[j]) + 1)]
                    s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(i, j, mk), zero_idxs=zero or None)
                    assert res["2"][0][k] == np.float32(s_or) and np.array_equal(res["2"][1][k], p_or), (kind, mode, i, j)
        arena.close()


def test_two_pass_forward_on_the_scores_kernel(nat, bba, monkeypatch):
    """PRALINE_TB_KEEP=1: the forward fill of the two-pass scheme runs on the staged scores kernel (k_dp_split16<..., KEEP>:
    H recurrence, LDS-DMA operand stream, shared-wave workgroups) and keeps the three states the recompute kernel starts
    from.  Identical scores, end states and paths as the single pass and as the oracle on the device's match scores:
    lengths around the 32-row checkpoint blocks and the 32-column strips, tasks with one wave and with shared waves,
    corner cells whose maximum is a gap state (very different lengths)."""
    rng = np.random.default_rng(43)
    for lens in ([1, 31, 32, 33, 64, 65, 97, 130, 200, 47, 85, 86, 300], [390, 412, 7, 640, 96, 128, 129, 1000]):
        profs = [synth_profile(rng, L)[0] for L in lens]
        pairs = np.array([(i, j) for i in range(len(lens)) for j in range(len(lens)) if i != j], dtype=np.int32)
        arena = nat.Arena(profs, bba["S"])
        res = {}
        for keep in ("0", "1"):
            monkeypatch.setenv("PRALINE_TB_KEEP", keep)
            monkeypatch.setenv("PRALINE_TB_TWOPASS", "0" if keep == "0" else "-1")
            plan = nat.Plan(arena, pairs, want_paths=True)
            mk = plan.match_kind()
            plan.run("global", *GAPS)
            res[keep] = (plan.scores().copy(), [p.copy() for p in plan.paths()], plan.kernel_name())
            plan.close()
        if "k_dp_split16<" not in res["1"][2] and any(os.environ.get(k) for k in ("PRALINE_KERNEL", "PRALINE_MM", "PRALINE_NO_STAGE", "PRALINE_NO_W2")):
            pytest.skip("the kept-state forward fill needs the staged scores kernel (switched off by the environment)")
        assert "k_dp_split16<" in res["1"][2] and res["1"][2].endswith("true>"), res["1"][2]
        assert np.array_equal(bits(res["0"][0]), bits(res["1"][0]))
        for k, (i, j) in enumerate(pairs):
            assert np.array_equal(res["0"][1][k], res["1"][1][k]), (i, j)
        for k in range(0, len(pairs), 5):
            i, j = pairs[k]
            s_or, p_or = oracle_dp_on_m("global", arena.match_scores(i, j, mk))
            assert res["1"][0][k] == np.float32(s_or) and np.array_equal(res["1"][1][k], p_or), (i, j)
        arena.close()


@pytest.mark.parametrize("lists", ["triangle", "ordered", "subset"])
def test_pipeline_workgroups_agree_bitwise(nat, bba, lists, monkeypatch):
    """k_dp_pipe (dp_pipe.hip.h: four waves pipeline the strips of tasks that share a set of 32 sequences one; shared
    operand ring, boundary hand-off through LDS) computes every cell with k_dp_split16's instructions: scores of float
    profiles must equal the task schedule's bit for bit in all five modes - ragged lengths (1 .. 2 strips up to several
    rounds, sets with empty lanes, tasks that start in the middle of a round, PRALINE_PIPE_BLOCK / SLOTS cutting the
    items differently) - and the oracle DP on the device's own match scores for a sample of pairs."""
    rng = np.random.default_rng({"triangle": 21, "ordered": 22, "subset": 23}[lists])
    N = 90
    lens = synth_lengths(rng, N, 140)
    lens[:8] = (1, 5, 31, 32, 33, 64, 65, 300)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    if lists == "triangle":
        pairs = all_pairs(N)
    else:
        pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
        if lists == "subset":
            pairs = pairs[rng.random(len(pairs)) < 0.9]
    for k in ("PRALINE_NO_PIPE", "PRALINE_PIPE_MIN_TASKS", "PRALINE_PIPE_BLOCK", "PRALINE_PIPE_SLOTS"):
        monkeypatch.delenv(k, raising=False)
    arena = nat.Arena(profs, bba["S"])
    if arena.info()["f16_terms"] not in (2, 3) or any(os.environ.get(k) for k in ("PRALINE_KERNEL", "PRALINE_MM", "PRALINE_NO_STAGE")):
        arena.close()
        pytest.skip("the pipeline workgroups run the float-profile instances of the staged stream")
    monkeypatch.setenv("PRALINE_NO_PIPE", "1")
    plan = nat.Plan(arena, pairs)
    want = {}
    for mode in MODES:
        plan.run(mode, *GAPS)
        want[mode] = plan.scores().copy()
    assert "k_dp_pipe" not in plan.kernel_name()
    kind = plan.match_kind()
    plan.close()
    monkeypatch.delenv("PRALINE_NO_PIPE")
    monkeypatch.setenv("PRALINE_PIPE_MIN_TASKS", "1")
    for env in ({}, {"PRALINE_PIPE_BLOCK": "3", "PRALINE_PIPE_SLOTS": "16"}, {"PRALINE_PIPE_BLOCK": "64", "PRALINE_PIPE_SLOTS": "100000"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = nat.Plan(arena, pairs)
        for mode in MODES:
            plan.run(mode, *GAPS)
            got = plan.scores()
            # (a cut that leaves too many lanes empty keeps the task schedule: only the default cut must take the pipeline)
            assert "k_dp_pipe" in plan.kernel_name() or env, plan.kernel_name()
            assert np.array_equal(bits(got), bits(want[mode])), (lists, env, mode, int((bits(got) != bits(want[mode])).sum()))
        # other gap scores on the same plan (the analytic column is rewritten by every launch)
        plan.run("global", -3.5, -0.25)
        got = plan.scores().copy()
        plan.close()
        for k in env:
            monkeypatch.delenv(k)
    for p in rng.choice(len(pairs), 12, replace=False):
        i, j = pairs[p]
        for mode in MODES:
            s_or, _ = oracle_dp_on_m(mode, arena.match_scores(int(i), int(j), kind))
            assert want[mode][p] == np.float32(s_or), (mode, i, j)
        s_or, _ = oracle_dp_on_m("global", arena.match_scores(int(i), int(j), kind), gaps=(-3.5, -0.25))
        assert got[p] == np.float32(s_or), (i, j)
    arena.close()


def test_prepared_schedule_gives_the_same_plan(nat, bba):
    """praline_sched_prepare / praline_plan_create_prepared: a score plan whose host scheduling ran ahead of the arena
    (on a worker thread: native.prepare_schedule_async) is the plan praline_plan_create builds - same kernel, same
    scores bit for bit; a schedule that does not fit the arena (other lengths, another pair list, an arena whose plans
    run another kernel) is ignored, misuse is refused."""
    rng = np.random.default_rng(41)
    N = 160
    lens = synth_lengths(rng, N, 120)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    pairs = all_pairs(N)
    arena = nat.Arena(profs, bba["S"])
    plan = nat.Plan(arena, pairs)
    plan.run("global", *GAPS)
    want, kernel = plan.scores().copy(), plan.kernel_name()
    plan.close()
    for how in ("sync", "async"):
        prep = nat.PreparedSchedule(lens, pairs) if how == "sync" else nat.prepare_schedule_async(lens, pairs)
        arena2 = nat.Arena(profs, bba["S"])
        plan = nat.Plan(arena2, pairs, prepared=prep)
        plan.run("global", *GAPS)
        assert plan.kernel_name() == kernel and np.array_equal(bits(plan.scores()), bits(want)), how
        plan.run("local", *GAPS)
        plan.close(); arena2.close()
    # schedules that do not belong to the arena / pair list are not used
    shuffled = pairs.copy()
    shuffled[1:-1] = shuffled[1:-1][rng.permutation(len(pairs) - 2)]          # same first / last pair, same count
    for bad in (nat.PreparedSchedule(lens[::-1].copy(), pairs), nat.PreparedSchedule(lens, pairs[::-1].copy()),
                nat.PreparedSchedule(lens, pairs[:-1]), nat.PreparedSchedule(lens, shuffled)):
        plan = nat.Plan(arena, pairs, prepared=bad)
        plan.run("global", *GAPS)
        assert np.array_equal(bits(plan.scores()), bits(want))
        plan.close()
    # an arena of plain sequences runs the lookup kernels: the prepared pipeline schedule is dropped
    onehot = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    a3 = nat.Arena(onehot, bba["S"])
    p_ref = nat.Plan(a3, pairs); p_ref.run("global", *GAPS)
    p_pre = nat.Plan(a3, pairs, prepared=nat.PreparedSchedule(lens, pairs)); p_pre.run("global", *GAPS)
    assert p_pre.kernel_name() == p_ref.kernel_name() and np.array_equal(bits(p_pre.scores()), bits(p_ref.scores()))
    p_ref.close(); p_pre.close(); a3.close()
    with pytest.raises(nat.NativeError):
        nat.PreparedSchedule(lens, np.array([[0, N]], dtype=np.int32))          # pair out of range
    with pytest.raises(nat.NativeError):
        nat.PreparedSchedule(np.array([5, 0, 3], dtype=np.int32), np.array([[0, 2]], dtype=np.int32))   # empty sequence
    with pytest.raises(ValueError):
        nat.Plan(arena, pairs, want_paths=True, prepared=nat.PreparedSchedule(lens, pairs))
    arena.close()


def test_float_profiles_with_more_than_21_active_symbols(nat):
    """Float profiles with mass on all 27 symbols and a score matrix without empty rows: 27 active symbols - past the 21
    that the K-packed three-term layout holds - so the hi / lo pieces of P and Q = P . S^T are laid out unpacked (six
    MFMAs per step).  The device's match scores agree with float64 to 1e-5 of the matrix's range (the north star's
    float tolerance), and the plans' scores and paths are the oracle's on those match scores."""
    rng = np.random.default_rng(57)
    A = 27
    S = rng.integers(-4, 12, (A, A)).astype(np.float32)
    S = np.maximum(S, S.T)
    np.fill_diagonal(S, rng.integers(4, 12, A))
    N = 40
    profs = []
    for L in rng.integers(50, 121, N):
        c = rng.random((int(L), A)).astype(np.float32) ** 3 + np.eye(A, dtype=np.float32)[rng.integers(0, A, int(L))]
        profs.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    arena = nat.Arena(profs, S)
    info = arena.info()
    assert info["n_active"] == 27 and info["f16_terms"] == 3, info
    pairs = all_pairs(N)
    plan = nat.Plan(arena, pairs, want_paths=True)
    kind = plan.match_kind()
    for (i, j) in ((0, 1), (3, 17), (38, 39), (11, 5)):
        m = arena.match_scores(i, j, kind)
        want = profs[i].astype(np.float64) @ S.astype(np.float64) @ profs[j].astype(np.float64).T
        assert np.abs(m - want).max() <= 1e-5 * np.abs(want).max(), (i, j, float(np.abs(m - want).max()))
    for mode in MODES:
        plan.run(mode, *GAPS)
        sc, paths = plan.scores().copy(), plan.paths()
        for p in rng.choice(len(pairs), 5, replace=False):
            i, j = pairs[p]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(int(i), int(j), kind))
            assert sc[p] == np.float32(s_or) and np.array_equal(paths[p], p_or), (mode, i, j)
    plan.close()
    plan = nat.Plan(arena, pairs)
    plan.run("global", *GAPS)
    s2 = plan.scores().copy()
    plan.close()
    plan = nat.Plan(arena, pairs, want_paths=True)
    plan.run("global", *GAPS)
    assert np.array_equal(bits(s2), bits(plan.scores()))      # scores-only and path kernels agree
    plan.close()
    arena.close()
