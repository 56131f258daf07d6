"""GPU tests of the reference-order match-score mode (PRALINE_MATCH_REFERENCE) and of float-profile ALIGNMENT parity
at BASELINE C2 scale: the checker is the CPU oracle fed with the REFERENCE-order match scores (never the device's)."""
import fractions
import json
import os

import numpy as np
import pytest

from conftest import MODES, ROOT, load_golden, one_hot, synth_lengths, synth_profile
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GO, GE = -11.0, -1.0


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    yield native
    native.set_match_mode(None)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def host_threads():
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        return 8


def run_device(nat, arena, pairs, modes, rects=None, want_paths=True):
    out = {}
    for mode in modes:
        plan = nat.Plan(arena, pairs, want_paths=want_paths, rects=rects)
        kind = plan.match_kind()
        plan.run(mode, GO, GE)
        sc = plan.scores()
        paths = [p.copy() for p in plan.paths()] if want_paths else None
        plan.close()
        out[mode] = (sc, paths, kind)
    return out


def test_reference_order_bitexact_small(nat, bba):
    """ref mode on the reference's own float-profile goldens (profile_profile.npz, produced by the real reference):
    scores bit-identical, paths identical, all modes; scores-only plans give the same bits."""
    d = load_golden("profile_profile.npz")
    profs = [d["profile%d" % i] for i in range(5)]
    pairs = np.array([(i, j) for i in range(5) for j in range(i + 1, 5)], dtype=np.int32)
    nat.set_match_mode("ref")
    try:
        arena = nat.Arena(profs, bba["S"])
        res = run_device(nat, arena, pairs, MODES)
        res0 = run_device(nat, arena, pairs, MODES, want_paths=False)
        arena.close()
    finally:
        nat.set_match_mode(None)
    for mode in MODES:
        sc, paths, kind = res[mode]
        assert kind == 2
        for k, (i, j) in enumerate(pairs):
            assert sc[k] == np.float32(d["score_%d_%d_%s" % (i, j, mode)]), (mode, i, j)
            assert np.array_equal(paths[k], d["path_%d_%d_%s" % (i, j, mode)]), (mode, i, j)
        assert np.array_equal(bits(res0[mode][0]), bits(sc)), mode


def test_reference_order_multiset_and_masks(nat, bba):
    """Several float track sets (one running sum per set, added in list order, cext.c:389-420) and Waterman-Eggert
    rectangles in ref mode: bit-identical to the oracle's reference-order evaluation."""
    rng = np.random.default_rng(23)
    lens = [41, 70, 33, 64, 9]
    p27 = [synth_profile(rng, L)[0] for L in lens]
    p3 = []
    for L in lens:
        c = rng.integers(0, 4, (L, 3)).astype(np.float32) + np.float32(0.25)
        p3.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    S3 = rng.normal(0, 2, (3, 3)).astype(np.float32)
    S = np.zeros((30, 30), dtype=np.float32)
    S[:27, :27] = bba["S"]
    S[27:, 27:] = S3
    cat = [np.concatenate([a, b], axis=1) for a, b in zip(p27, p3)]
    pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j], dtype=np.int32)
    rects = [[(3, 8, 2, 9)] if k % 2 else [] for k in range(len(pairs))]
    nat.set_match_mode("ref")
    try:
        arena = nat.Arena(cat, S, set_sizes=[27, 3])
        res = run_device(nat, arena, pairs, MODES)
        res_l = run_device(nat, arena, pairs, ["local"], rects=rects)
        arena.close()
    finally:
        nat.set_match_mode(None)
    for k, (i, j) in enumerate(pairs):
        for mode in MODES:
            s_or, p_or = orc.pairwise_align(mode, [p27[i], p3[i]], [p27[j], p3[j]], [bba["S"], S3], (GO, GE))
            assert res[mode][0][k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(res[mode][1][k], p_or), (mode, i, j)
        zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)
                if y <= lens[i] and x <= lens[j]]
        s_or, p_or = orc.pairwise_align("local", [p27[i], p3[i]], [p27[j], p3[j]], [bba["S"], S3], (GO, GE),
                                        zero_idxs=zero or None)
        assert res_l["local"][0][k] == np.float32(s_or), (i, j)
        assert np.array_equal(res_l["local"][1][k], p_or), (i, j)


def test_reference_order_tiles(nat, bba, monkeypatch):
    """Reference-order plans on arenas of up to 32 symbols and 8 nonzeros per row run k_match_tile (packed fp32, table rows
    in LDS) + the dense-tile instances of the split-strip kernels (dp_reftile.hip.h).  Ragged lengths around the strip /
    chunk / row-group boundaries, two track sets, Waterman-Eggert rectangles, every mode, scores-only and with paths:
    bit-identical to the oracle's reference-order evaluation (cext.c:33-97,389-420) AND to the one-cell-per-thread
    kernels' tiles (PRALINE_NO_REFTILE=1); a second run of the same plan reuses stale tile memory.  With a tile budget of 2 MiB
    the longer tasks are swept a range of strips per launch (boundary column and local maximum carried between launches)."""
    rng = np.random.default_rng(41)
    lens = [41, 70, 33, 64, 9, 130, 257, 1, 2, 31, 32, 33, 127, 128, 129, 300, 16, 17, 15]
    p27 = [synth_profile(rng, L)[0] for L in lens]
    assert max(int((p != 0).sum(axis=1).max()) for p in p27) <= 7
    p3 = []
    for L in lens:
        c = np.zeros((L, 3), dtype=np.float32)          # one nonzero per row: 8 per row with the first set's 7
        c[np.arange(L), rng.integers(0, 3, L)] = rng.uniform(0.5, 1.5, L).astype(np.float32)
        p3.append(c)
    S3 = rng.normal(0, 2, (3, 3)).astype(np.float32)
    S2 = np.zeros((30, 30), dtype=np.float32)
    S2[:27, :27] = bba["S"]
    S2[27:, 27:] = S3
    cat = [np.concatenate([a, b], axis=1) for a, b in zip(p27, p3)]
    n = len(lens)
    pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
    rects = [[(3, 8, 2, 9), (20, 30, 25, 40)] if k % 3 == 0 else [] for k in range(len(pairs))]
    check = np.random.default_rng(3).choice(len(pairs), 60, replace=False)      # pairs compared with the (slow) oracle

    def run(arena, mode, want_paths, tile, rects=None, twice=False):
        monkeypatch.setenv("PRALINE_NO_REFTILE", "0" if tile else "1")
        plan = nat.Plan(arena, pairs, want_paths=want_paths, rects=rects)
        plan.run(mode, GO, GE)
        if twice:
            plan.run("local" if mode != "local" else "global", GO, GE)       # leaves other scores in the tile memory
            plan.run(mode, GO, GE)
        assert plan.tile_producer() == (1 if tile else 2)
        out = (plan.scores().copy(), [p.copy() for p in plan.paths()] if want_paths else None, plan.kernel_name(), plan.match_kind())
        plan.close()
        return out

    nat.set_match_mode("ref")
    try:
        for name, profs, S, sets in (("one set", p27, bba["S"], None), ("two sets", cat, S2, [27, 3])):
            arena = nat.Arena(profs, S, set_sizes=sets) if sets else nat.Arena(profs, S)
            for mode in MODES:
                t_sc, _, t_kn, t_kind = run(arena, mode, False, True, twice=True)
                o_sc, _, o_kn, _ = run(arena, mode, False, False)
                # (dense-tile instance, BSRC = 4: one-task waves or - small one-chunk plans - shared-wave workgroups)
                assert "k_dp_split16<1, 1" in t_kn and (", 4, 1, false>" in t_kn or ", 4, 4, false>" in t_kn) and "k_dp_split16<1, 1" in o_kn, (t_kn, o_kn)
                assert t_kind == 2
                assert np.array_equal(bits(t_sc), bits(o_sc)), (name, mode)
                monkeypatch.setenv("PRALINE_NO_W2", "1")                      # one-task waves instead of shared-wave workgroups
                w_sc, _, w_kn, _ = run(arena, mode, False, True)
                monkeypatch.delenv("PRALINE_NO_W2")
                assert ", 4, 1, false>" in w_kn and np.array_equal(bits(w_sc), bits(t_sc)), (name, mode, w_kn)
                t_sc2, t_paths, t_kn, _ = run(arena, mode, True, True)
                o_sc2, o_paths, _, _ = run(arena, mode, True, False)
                assert "k_dp_split16_tb<1, 3" in t_kn and t_kn.endswith(", 4, false, false>"), t_kn
                assert np.array_equal(bits(t_sc2), bits(o_sc2)) and np.array_equal(bits(t_sc2), bits(t_sc)), (name, mode)
                assert all(np.array_equal(x, y) for x, y in zip(t_paths, o_paths)), (name, mode)
                for k in check:
                    i, j = pairs[k]
                    tracks = ([p27[i], p3[i]], [p27[j], p3[j]], [bba["S"], S3]) if sets else ([p27[i]], [p27[j]], [bba["S"]])
                    s_or, p_or = orc.pairwise_align(mode, tracks[0], tracks[1], tracks[2], (GO, GE))
                    assert t_sc2[k] == np.float32(s_or) and np.array_equal(t_paths[k], p_or), (name, mode, i, j)
            # several launch chunks (tile budget of 2 MiB: a few tasks per chunk), scores-only and with paths
            monkeypatch.setenv("PRALINE_REFTILE_BUDGET_MB", "2")
            c_sc, _, _, _ = run(arena, "semiglobal_one", False, True)
            c_sc2, c_paths, _, _ = run(arena, "local", True, True)
            monkeypatch.delenv("PRALINE_REFTILE_BUDGET_MB")
            o_sc, _, _, _ = run(arena, "semiglobal_one", False, False)
            o_sc2, o_paths, _, _ = run(arena, "local", True, False)
            assert np.array_equal(bits(c_sc), bits(o_sc)) and np.array_equal(bits(c_sc2), bits(o_sc2)), name
            assert all(np.array_equal(x, y) for x, y in zip(c_paths, o_paths)), name
            # Waterman-Eggert rectangles (local; the register-resident masks of the split-strip path kernels)
            t_sc, t_paths, t_kn, _ = run(arena, "local", True, True, rects=rects)
            o_sc, o_paths, _, _ = run(arena, "local", True, False, rects=rects)
            assert "true, true, false, false, 4, false, false>" in t_kn, t_kn
            assert np.array_equal(bits(t_sc), bits(o_sc)) and all(np.array_equal(x, y) for x, y in zip(t_paths, o_paths)), name
            arena.close()
    finally:
        nat.set_match_mode(None)


def exact_path_score(path, p1, p2, S, mode, L1, L2):
    """Score of an alignment path in EXACT rational arithmetic on the float32 inputs: diagonal steps add
    sum_ij p1[y,i] S[i,j] p2[x,j]; a run of k gap steps costs GO + (k - 1) GE unless it runs along a free edge
    (semiglobal, praline/component/align.py:371-385,411-424)."""
    F = fractions.Fraction
    path = np.asarray(path)
    dy, dx = np.diff(path[:, 0]), np.diff(path[:, 1])
    kinds = np.where((dy == 1) & (dx == 1), 0, np.where(dy == 1, 1, 2))
    total = F(0)
    r, n = 0, len(kinds)
    while r < n:
        k = kinds[r]
        if k == 0:
            y, x = path[r + 1]
            a, b = p1[y - 1], p2[x - 1]
            for i in np.flatnonzero(a):
                for j in np.flatnonzero(b):
                    total += F(float(a[i])) * F(float(S[i, j])) * F(float(b[j]))
            r += 1
            continue
        e = r
        while e < n and kinds[e] == k:
            e += 1
        free = False
        if mode.startswith("semiglobal"):
            y0, x0 = path[r]
            if k == 1 and ((x0 == 0 and mode in ("semiglobal_both", "semiglobal_one")) or x0 == L2):
                free = True
            if k == 2 and (y0 == 0 or y0 == L1) and mode in ("semiglobal_both", "semiglobal_two"):
                free = True
        if not free:
            total += F(GO) + (e - r - 1) * F(GE)
        r = e
    return total


def test_c2_float_profile_alignments_vs_reference_order(nat, bba):
    """BASELINE C2's own float profiles (256 seqs ~400 aa, seed 2) x 5 modes with paths, checked against the oracle on the
    REFERENCE-order match scores:
      * ref mode, ALL 32 640 pairs: every score bit-identical and every path identical to the reference's;
      * default (matrix-pipe) mode, 2 048 sampled pairs: every score within 1e-5 relative; a path may differ only where the choice is a tie
        up to float32 rounding - each differing path's score in EXACT rational arithmetic is within 1e-5 relative of
        the reference path's exact score.  The mismatch statistics are written to gpurun_out/ for profiles/."""
    rng = np.random.default_rng(2)
    N = 256
    lens = synth_lengths(rng, N, 400).astype(np.int32)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    S = bba["S"]
    allp = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
    sel = np.sort(np.random.default_rng(7).choice(len(allp), 2048, replace=False))
    pairs = allp[sel]
    cat = np.concatenate(profs, axis=0)
    row_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    sc_ref, paths_ref = orc.batch_align(MODES, cat, row_off, lens, S, pairs, GO, GE, threads=host_threads())

    # ref mode: ALL 32 640 pairs x 5 modes = 163 200 alignments (in chunks: the oracle's paths of a chunk are a few hundred
    # MB), every score bit for bit and every path row for row
    nat.set_match_mode("ref")
    n_strict = 0
    try:
        arena = nat.Arena(profs, S)
        for c0 in range(0, len(allp), 8192):
            chunk = allp[c0:c0 + 8192]
            sc_c, paths_c = orc.batch_align(MODES, cat, row_off, lens, S, chunk, GO, GE, threads=host_threads())
            strict = run_device(nat, arena, chunk, MODES)
            for q, mode in enumerate(MODES):
                sc, paths, kind = strict[mode]
                assert kind == 2
                assert np.array_equal(bits(sc), bits(sc_c[:, q])), (mode, c0)
                bad = [k for k in range(len(chunk)) if not np.array_equal(paths[k], paths_c[k][q])]
                assert not bad, (mode, c0, len(bad), chunk[bad[:3]].tolist())
                n_strict += len(chunk)
            del strict, sc_c, paths_c
        arena.close()
    finally:
        nat.set_match_mode(None)
    assert n_strict == len(allp) * len(MODES)

    arena = nat.Arena(profs, S)
    fast = run_device(nat, arena, pairs, MODES)
    arena.close()
    stats = {"workload": "C2 float profiles (seed 2), %d sampled pairs x %d modes" % (len(pairs), len(MODES)),
             "alignments": int(len(pairs) * len(MODES)), "modes": {}}
    worst_rel = 0.0
    for q, mode in enumerate(MODES):
        sc, paths, kind = fast[mode]
        assert kind == 1
        rel = np.abs(sc - sc_ref[:, q]) / np.maximum(1.0, np.abs(sc_ref[:, q]))
        assert rel.max() <= 1e-5, (mode, rel.max())
        worst_rel = max(worst_rel, float(rel.max()))
        diff = [k for k in range(len(pairs)) if not np.array_equal(paths[k], paths_ref[k][q])]
        exact_ties = 0
        for k in diff:
            i, j = pairs[k]
            e_dev = exact_path_score(paths[k], profs[i], profs[j], S, mode, lens[i], lens[j])
            e_ref = exact_path_score(paths_ref[k][q], profs[i], profs[j], S, mode, lens[i], lens[j])
            # both paths are optimal up to float32 rounding: neither beats the other by more than the tolerance
            assert abs(e_dev - e_ref) <= fractions.Fraction(1, 100000) * max(1, abs(e_ref)), (mode, i, j, float(e_dev), float(e_ref))
            exact_ties += e_dev == e_ref
        stats["modes"][mode] = {"paths_differing": len(diff), "of": int(len(pairs)), "exact_arithmetic_ties": int(exact_ties),
                                "max_rel_score_diff": float(rel.max())}
    stats["paths_differing_total"] = int(sum(v["paths_differing"] for v in stats["modes"].values()))
    stats["max_rel_score_diff"] = worst_rel
    stats["ref_mode"] = "all %d alignments of the WHOLE workload (%d pairs x %d modes): scores bit-identical, paths identical" % (
        n_strict, len(allp), len(MODES))
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_c2_float_paths.json"), "w") as f:
            json.dump(stats, f, indent=1)
    except OSError:
        pass
    print("C2 float-profile path parity:", json.dumps(stats))
    # the fast mode is allowed to differ only on rounding-level ties, and rarely: 1 of 10 240 alignments was measured
    # (profiles/r02_parity_c2_float_paths.json); ten times that is the bound
    assert stats["paths_differing_total"] <= 0.001 * stats["alignments"], stats


def test_wide_alphabets_take_the_reference_order_path(nat):
    """More than 32 active symbols (the reference has no such limit, cext.c:389-420): the arena keeps the raw profiles
    and its plans run the reference-order path - scores and paths bit-identical to the oracle, every mode, float
    profiles over a 60-symbol alphabet, with rectangles; praline_build_scores likewise."""
    rng = np.random.default_rng(31)
    A = 60
    S = rng.normal(0, 3, (A, A)).astype(np.float32)
    lens = [35, 51, 8, 40]
    profs = []
    for L in lens:
        c = np.zeros((L, A), dtype=np.float32)
        for _ in range(5):
            c[np.arange(L), rng.integers(0, A, L)] += rng.integers(1, 4, L)
        profs.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    arena = nat.Arena(profs, S)
    assert arena.info()["n_active"] > 32
    pairs = np.array([(i, j) for i in range(4) for j in range(4) if i != j], dtype=np.int32)
    rects = [[(2, 5, 1, 6)] if k % 2 else [] for k in range(len(pairs))]
    res = run_device(nat, arena, pairs, MODES)
    res_l = run_device(nat, arena, pairs, ["local"], rects=rects)
    res0 = run_device(nat, arena, pairs, ["global"], want_paths=False)
    for k, (i, j) in enumerate(pairs):
        for mode in MODES:
            assert res[mode][2] == 2
            s_or, p_or = orc.pairwise_align(mode, [profs[i]], [profs[j]], [S], (GO, GE))
            assert res[mode][0][k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(res[mode][1][k], p_or), (mode, i, j)
        zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)
                if y <= lens[i] and x <= lens[j]]
        s_or, p_or = orc.pairwise_align("local", [profs[i]], [profs[j]], [S], (GO, GE), zero_idxs=zero or None)
        assert res_l["local"][0][k] == np.float32(s_or) and np.array_equal(res_l["local"][1][k], p_or), (i, j)
    assert np.array_equal(bits(res0["global"][0]), bits(res["global"][0]))
    m_dev = arena.match_scores(0, 1, kind=2)
    m_ref = np.zeros_like(m_dev)
    orc.cext_build_scores([profs[0]], [profs[1]], [orc.build_nonzero_matrix(profs[0])], [orc.build_nonzero_matrix(profs[1])], [S], m_ref)
    assert np.array_equal(bits(m_dev), bits(m_ref))
    arena.close()
    m = np.zeros((lens[0], lens[1]), dtype=np.float32)
    nat.cext_build_scores([profs[0]], [profs[1]], None, None, [S], m)       # the drop-in twin on a wide alphabet
    assert np.array_equal(bits(m), bits(m_ref))


def test_many_rectangles_per_pair(nat, bba):
    """More zero rectangles per pair than the split-strip kernels hold in registers (Waterman-Eggert with many
    iterations, preprofile.py:247-255): one batched submission on the reference-order path with per-row column masks;
    scores and paths bit-identical to the oracle's, one-hot and float profiles."""
    rng = np.random.default_rng(37)
    lens = [60, 75, 48, 66, 90]
    onehots = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
    floats = [synth_profile(rng, L)[0] for L in lens]
    pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j], dtype=np.int32)
    rects = []
    for k, (i, j) in enumerate(pairs):
        n = [0, 3, 7, 12][k % 4]
        rl = []
        for _ in range(n):
            y0, x0 = int(rng.integers(1, lens[i])), int(rng.integers(1, lens[j]))
            rl.append((y0, min(lens[i], y0 + int(rng.integers(0, 9))), x0, min(lens[j], x0 + int(rng.integers(0, 9)))))
        rects.append(rl)
    assert max(len(r) for r in rects) > nat.MAX_RECTS
    for profs in (onehots, floats):
        arena = nat.Arena(profs, bba["S"])
        for mode in MODES:
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
            kind = plan.match_kind()
            plan.run(mode, GO, GE)
            sc, paths = plan.scores(), plan.paths()
            plan.close()
            for k, (i, j) in enumerate(pairs):
                zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
                # float profiles: such plans evaluate the match scores in the reference's order (experiment builds:
                # scripts/exp_mask2.py); plain sequences stay on k_dp_quad_tb, which reads per-row mask words (integer
                # scoring: the same scores in any order)
                assert kind == 2 or profs is onehots or os.environ.get("PRALINE_EXP_BATCH_MASK2")
                s_or, p_or = orc.pairwise_align(mode, [profs[i]], [profs[j]], [bba["S"]], (GO, GE), zero_idxs=zero or None)
                assert sc[k] == np.float32(s_or), (mode, i, j, len(rects[k]))
                assert np.array_equal(paths[k], p_or), (mode, i, j, len(rects[k]))
        arena.close()
