"""GPU tests at BASELINE.json's full configuration sizes (all of C3 incl. the second Waterman-Eggert pass, one rank's
share of C4, all of C5): properties that do not need the oracle at full size - the score recomputed from the returned path, mode ordering, symmetry
and self-alignment under integer scoring - plus oracle spot checks on sampled pairs."""
import os

import numpy as np
import pytest

from conftest import load_golden, one_hot, synth_lengths, synth_profile
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GO, GE = -11.0, -1.0
MODES5 = ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two")
GAPS = (GO, GE)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_dp_on_m(mode, m):
    """The oracle's fill + end cell + traceback on a given match-score matrix (the device's own m)."""
    g1, g2 = orc.gap_arrays(m.shape[0], m.shape[1], GAPS)
    return orc.raw_pairwise_align(mode, np.ascontiguousarray(m), g1, g2, None)


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    return native


def path_score(path, v1, v2, S, mode):
    """Score of an alignment path under affine gaps open GO / extend GE (praline/util/cext.c:152-283):
    diagonal steps add S[a, b]; a run of k gap steps costs GO + (k - 1) GE.  Semiglobal paths come extended
    to the corners (util/align.py:268-297): runs along a free edge are not charged."""
    path = np.asarray(path)
    dy = np.diff(path[:, 0])
    dx = np.diff(path[:, 1])
    assert np.all((dy >= 0) & (dx >= 0) & (dy + dx >= 1) & (dy <= 1) & (dx <= 1)), "not a monotone unit-step path"
    L1, L2 = len(v1), len(v2)
    kinds = np.where((dy == 1) & (dx == 1), 0, np.where(dy == 1, 1, 2))     # 0 match, 1 up (gap in two), 2 left
    total = 0.0
    r = 0
    n = len(kinds)
    while r < n:
        k = kinds[r]
        if k == 0:
            total += float(S[v1[path[r + 1, 0] - 1], v2[path[r + 1, 1] - 1]])
            r += 1
            continue
        e = r
        while e < n and kinds[e] == k:
            e += 1
        free = False
        if mode.startswith("semiglobal"):
            y0, x0 = path[r]
            y1, x1 = path[e]
            # start: column 0 is free for both/one, row 0 for both/two (align.py:371-385); end: the last
            # column may hold the end cell in every semiglobal mode, the last row only for both/two
            # (align.py:411-424)
            if k == 1 and ((x0 == 0 and mode in ("semiglobal_both", "semiglobal_one")) or x0 == L2):
                free = True
            if k == 2 and (y0 == 0 or y0 == L1) and mode in ("semiglobal_both", "semiglobal_two"):
                free = True
        if not free:
            total += GO + (e - r - 1) * GE
        r = e
    return total


def host_threads():
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        return 8


def test_c3_preprofile_modes_full_size(nat, bba):
    """BASELINE config 2 at its stated size: 1024 seqs ~250 aa, ALL 1 047 552 ordered (master, slave) pairs with paths.
    Local mode runs both Waterman-Eggert passes (the second one masking the bounding box of each first-pass path,
    praline/component/preprofile.py:227-267); global and the three semiglobal modes run one pass.  Checks: the paths'
    recomputed scores, the mode ordering, oracle scores + paths (with rectangles) on sampled pairs."""
    rng = np.random.default_rng(3)
    N = 1024
    lens = synth_lengths(rng, N, 250).astype(np.int32)
    vals = [rng.integers(0, 20, int(L)) for L in lens]
    S = bba["S"]
    profs = [one_hot(v, 27) for v in vals]
    arena = nat.Arena(profs, S)
    ii, jj = np.divmod(np.arange(N * N, dtype=np.int64), N)
    allp = np.stack([ii[ii != jj], jj[ii != jj]], axis=1).astype(np.int32)
    assert len(allp) == 1047552
    cat = np.concatenate(profs, axis=0)
    row_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    spot = np.sort(rng.choice(len(allp), 48, replace=False))
    results = {}
    for mode in ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"):
        plan = nat.Plan(arena, allp, want_paths=True)
        plan.run(mode, GO, GE)
        sc = plan.scores()
        buf, off, rows = plan.paths_packed()
        bounds = plan.path_bounds() if mode == "local" else None
        plan.close()
        results[mode] = sc
        assert np.isfinite(sc).all()
        for k in rng.integers(0, len(allp), 200):
            i, j = allp[k]
            path = buf[off[k]:off[k] + rows[k]]
            if mode != "local":
                assert tuple(path[0]) == (0, 0) and tuple(path[-1]) == (lens[i], lens[j]), (mode, i, j)
            assert path_score(path, vals[i], vals[j], S, mode) == sc[k], (mode, i, j)
        sc_or, p_or = orc.batch_align([mode], cat, row_off, lens, S, allp[spot], GO, GE, threads=host_threads())
        for q, k in enumerate(spot):
            assert sc[k] == sc_or[q, 0], (mode, allp[k])
            assert np.array_equal(buf[off[k]:off[k] + rows[k]], p_or[q][0]), (mode, allp[k])
        if mode == "local":
            # second Waterman-Eggert pass over the full list: every pair masks its own first-pass bounding box
            for k in spot:
                pth = buf[off[k]:off[k] + rows[k]]
                assert tuple(bounds[k]) == (pth[:, 0].min(), pth[:, 0].max(), pth[:, 1].min(), pth[:, 1].max())
            del buf
            plan = nat.Plan(arena, allp, want_paths=True, rects=bounds.reshape(-1, 1, 4))
            plan.run("local", GO, GE)
            sc2 = plan.scores()
            buf2, off2, rows2 = plan.paths_packed()
            plan.close()
            assert np.isfinite(sc2).all() and (sc2 <= sc).all()      # masking cells can only lower the best local score
            rect_list = [[tuple(int(v) for v in bounds[k])] for k in spot]
            sc_or2, p_or2 = orc.batch_align(["local"], cat, row_off, lens, S, allp[spot], GO, GE, rects=rect_list,
                                            threads=host_threads())
            for q, k in enumerate(spot):
                assert sc2[k] == sc_or2[q, 0], allp[k]
                assert np.array_equal(buf2[off2[k]:off2[k] + rows2[k]], p_or2[q][0]), allp[k]
                b = bounds[k]
                pth = buf2[off2[k]:off2[k] + rows2[k]]
                # no aligned cell of the second path lies inside the masked rectangle
                inside = (pth[1:, 0] >= b[0]) & (pth[1:, 0] <= b[1]) & (pth[1:, 1] >= b[2]) & (pth[1:, 1] <= b[3])
                assert not inside.any(), allp[k]
            del buf2
    # (local >= semiglobal_both does NOT hold in the reference: local mode keeps the penalised global boundary,
    # align.py:371-385, so an alignment that starts in the first row or column scores less than with free ends)
    for hi, lo in (("semiglobal_both", "semiglobal_one"), ("semiglobal_both", "semiglobal_two"),
                   ("semiglobal_one", "global"), ("semiglobal_two", "global")):
        bad = np.nonzero(~(results[hi] >= results[lo]))[0]
        assert len(bad) == 0, (hi, lo, len(bad), allp[bad[:3]].tolist(), results[hi][bad[:3]], results[lo][bad[:3]])
    arena.close()


def test_c3_build_preprofiles_full_size_two_pass_local(nat, bba):
    """The product call on all of C3: build_preprofiles(mode="local", waterman_eggert_iterations=2) - 2 x 1 047 552
    alignments with paths, counted on the device - against the operator chain the reference runs per master
    (LocalMasterSlaveAligner + ProfileBuilder, preprofile.py:213-269, profile.py:41-74) for sampled masters."""
    from praline_amd import component as comp
    from praline_amd import container as ct
    from praline_amd import core
    rng = np.random.default_rng(3)
    N = 1024
    lens = synth_lengths(rng, N, 250).astype(np.int32)
    vals = [rng.integers(0, 20, int(L)) for L in lens]
    seqs = [ct.Sequence("s%04d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
            for i, v in enumerate(vals)]
    blosum = ct.blosum62()
    tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode="local", waterman_eggert_iterations=2)
    assert len(tracks) == N
    total = sum(int(np.asarray(t.counts).sum()) for t in tracks)
    assert total > int(lens.sum())      # the slaves contributed
    idx = core.TypeIndex()
    idx.autoregister()
    manager = core.Manager(idx)
    for i in (0, 517, 1023):
        ex = core.Execution(manager, "root")
        ex.add_task(comp.LocalMasterSlaveAligner).environment(core.Environment({}), core.Environment({})).inputs(
            master_sequence=seqs[i], slave_sequences=[s for k, s in enumerate(seqs) if k != i],
            track_id_sets=[[ct.TRACK_ID_INPUT]], score_matrices=[blosum])
        aln = core.run(ex)[0]['alignment']
        ex = core.Execution(manager, "root")
        ex.add_task(comp.ProfileBuilder).environment(core.Environment({}), core.Environment({})).inputs(
            alignment=aln, track_id=ct.TRACK_ID_INPUT)
        want = core.run(ex)[0]['profile_track']
        assert np.array_equal(np.asarray(tracks[i].counts), np.asarray(want.counts)), i


def test_c4_rank_share_full_size(nat, bba, monkeypatch):
    """BASELINE config 3: 4096 seqs ~400 aa over 8 ranks - one rank's column shard (1.05 M pairs, 1.7e11
    cells), float profiles and one-hot: finite scores, exact symmetry and self-alignment under integer
    scoring, oracle spot checks."""
    from praline_amd import allpairs
    rng = np.random.default_rng(4)
    N = 4096
    lens = synth_lengths(rng, N, 400)
    pairs = allpairs.enumerate_pairs(N)
    shard = allpairs.shard_columns(lens, pairs, 8)[3]
    mine = pairs[shard]
    assert abs(len(mine) - len(pairs) / 8) < 0.05 * len(pairs) / 8
    S = bba["S"]
    vals = [rng.integers(0, 20, int(L)) for L in lens]
    arena = nat.Arena([one_hot(v, 27) for v in vals], S)
    plan = nat.Plan(arena, mine)
    plan.run("global", GO, GE)
    sc = plan.scores()
    plan.close()
    assert np.isfinite(sc).all()
    sub = rng.integers(0, len(mine), 4000)
    plan = nat.Plan(arena, mine[sub][:, ::-1].copy())
    plan.run("global", GO, GE)
    assert np.array_equal(plan.scores(), sc[sub])          # integer scoring: score(i, j) == score(j, i) exactly
    plan.close()
    diag = np.array([(i, i) for i in range(0, N, 64)], dtype=np.int32)
    plan = nat.Plan(arena, diag)
    plan.run("global", GO, GE)
    want = np.array([S[vals[i], vals[i]].sum() for i in diag[:, 0]], dtype=np.float32)
    assert np.array_equal(plan.scores(), want)             # self alignment: the sum of the diagonal entries
    plan.close()
    for k in rng.integers(0, len(mine), 5):
        i, j = mine[k]
        assert sc[k] == np.float32(orc.pairwise_score_fast("global", one_hot(vals[i], 27), one_hot(vals[j], 27), S, GO, GE))
    # the other modes on the full shard (local runs the one-hot table kernel in four-wave workgroups)
    for mode in ("local", "semiglobal_both"):
        plan = nat.Plan(arena, mine)
        plan.run(mode, GO, GE)
        scm = plan.scores()
        plan.close()
        assert np.isfinite(scm).all() and (scm >= sc).all()      # both only free something the global path pays for
        for k in rng.integers(0, len(mine), 5):
            i, j = mine[k]
            assert scm[k] == np.float32(orc.pairwise_score_fast(mode, one_hot(vals[i], 27), one_hot(vals[j], 27), S, GO, GE)), mode
    arena.close()
    # float profiles on the WHOLE shard (1.05 M pairs: the pipeline workgroups, k_dp_pipe): 24 pairs against the oracle in
    # the reference's summation order (1e-5 relative, north_star) and bit for bit against the oracle DP on the device's own
    # match scores; a 40 000-pair sample bit for bit against the task schedule's kernel (k_dp_split16)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    arena = nat.Arena(profs, S)
    plan = nat.Plan(arena, mine)
    plan.run("global", GO, GE)
    scf = plan.scores().copy()
    kind, kernel = plan.match_kind(), plan.kernel_name()
    plan.close()
    assert np.isfinite(scf).all()
    for k in rng.choice(len(mine), 24, replace=False):
        i, j = mine[k]
        ref = orc.pairwise_score_fast("global", profs[i], profs[j], S, GO, GE)
        assert abs(scf[k] - ref) <= 1e-5 * abs(ref), (i, j)
        s_or, _ = oracle_dp_on_m("global", arena.match_scores(int(i), int(j), kind))
        assert scf[k] == np.float32(s_or), (i, j)
    if "k_dp_pipe" in kernel:
        monkeypatch.setenv("PRALINE_NO_PIPE", "1")
        sample = np.sort(rng.choice(len(mine), 40000, replace=False))
        plan = nat.Plan(arena, mine[sample])
        plan.run("global", GO, GE)
        assert "k_dp_pipe" not in plan.kernel_name()
        assert np.array_equal(bits(plan.scores()), bits(scf[sample]))
        plan.close()
        monkeypatch.delenv("PRALINE_NO_PIPE")
    arena.close()
    # the same shard in the reference-order mode (k_match_tile + dense-tile instances; ~700 GB of tiles in 32 GiB launch
    # chunks, groups of several hundred tasks): 16 pairs bit-identical to the oracle's reference-order evaluation, and a
    # 2 000-pair sample bit for bit against the tiles of the one-cell-per-thread kernels
    nat.set_match_mode("ref")
    try:
        arena = nat.Arena(profs, S)
        plan = nat.Plan(arena, mine)
        plan.run("global", GO, GE)
        scr = plan.scores().copy()
        assert plan.match_kind() == 2 and plan.tile_producer() == 1 and ", 4, " in plan.kernel_name(), plan.kernel_name()
        plan.close()
        for k in rng.choice(len(mine), 16, replace=False):
            i, j = mine[k]
            assert scr[k] == np.float32(orc.pairwise_score_fast("global", profs[i], profs[j], S, GO, GE)), (i, j)
        rel = np.abs(scr - scf) / np.maximum(1.0, np.abs(scf))
        assert rel.max() <= 1e-5, rel.max()                      # (and the default mode agrees with it to north_star's tolerance)
        monkeypatch.setenv("PRALINE_NO_REFTILE", "1")
        sample = np.sort(rng.choice(len(mine), 2000, replace=False))
        plan = nat.Plan(arena, mine[sample])
        plan.run("global", GO, GE)
        assert plan.tile_producer() == 2 and ", 4, " in plan.kernel_name(), plan.kernel_name()
        assert np.array_equal(bits(plan.scores()), bits(scr[sample]))
        plan.close()
        monkeypatch.delenv("PRALINE_NO_REFTILE")
        arena.close()
    finally:
        nat.set_match_mode(None)


@pytest.mark.parametrize("kind", ["float", "onehot"])
def test_c4_eight_column_shards_equal_one_plan(nat, bba, kind):
    """BASELINE config 4 split as `bench.py --gpus 8` splits it, short of the RCCL call: the eight column shards
    (allpairs.shard_columns) run one after the other on this GPU, their score slices are padded and lined up rank by rank as
    the all-gather lines them up, `out[dst] = gathered[src]` (allpairs.gather_maps) puts them back into the reference's
    row-major pair order (tree.py:105-129,142-145) - and the result equals, bit for bit, ONE plan over all 8 386 560 pairs.
    Float profiles (k_dp_pipe on every shard and on the whole list) and plain sequences (the lookup kernel)."""
    from praline_amd import allpairs
    rng = np.random.default_rng(4)
    N = 4096
    lens = synth_lengths(rng, N, 400)
    pairs = allpairs.enumerate_pairs(N)
    if kind == "float":
        profs = [synth_profile(rng, int(L))[0] for L in lens]
    else:
        profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    arena = nat.Arena(profs, bba["S"])
    del profs
    shards = allpairs.shard_columns(lens, pairs, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(len(pairs)))      # a partition of the pair list
    cells = lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]
    loads = np.array([cells[ix].sum() for ix in shards], dtype=np.float64)
    assert loads.max() / loads.mean() < 1.01                                        # balanced by DP cells
    src, dst, shard_len = allpairs.gather_maps(shards)
    gathered = np.full(8 * shard_len, np.nan, dtype=np.float32)
    kernels = set()
    for r, ix in enumerate(shards):
        plan = nat.Plan(arena, pairs[ix])
        plan.run("global", GO, GE)
        gathered[r * shard_len:r * shard_len + len(ix)] = plan.scores()
        kernels.add(plan.kernel_name())
        plan.close()
    ordered = np.empty(len(pairs), dtype=np.float32)
    ordered[dst] = gathered[src]
    plan = nat.Plan(arena, pairs)
    plan.run("global", GO, GE)
    whole = plan.scores().copy()
    kernels.add(plan.kernel_name())
    plan.close()
    arena.close()
    assert np.isfinite(whole).all()
    assert np.array_equal(bits(ordered), bits(whole)), kernels
    assert all(("k_dp_pipe" in k) == (kind == "float") for k in kernels), kernels


def test_c5_long_dna_full_size(nat):
    """BASELINE config 4 at its stated size: 512 nucleotide seqs ~5 kb, ALL 130 816 pairs (3.3e12 cells) in global,
    local and semiglobal_both mode; exact against the oracle on 24 sampled pairs per mode; symmetry of the scores under
    integer scoring; mode ordering."""
    from praline_amd import allpairs
    d = load_golden("synthetic_dna.npz")
    S = d["matrix"]
    rng = np.random.default_rng(5)
    N = 512
    lens = synth_lengths(rng, N, 5000).astype(np.int32)
    vals = [rng.integers(0, 4, int(L)) for L in lens]
    profs = [one_hot(v, 15) for v in vals]
    arena = nat.Arena(profs, S)
    pairs = allpairs.enumerate_pairs(N)
    assert len(pairs) == 130816
    modes = ("global", "local", "semiglobal_both")
    res = {}
    for mode in modes:
        plan = nat.Plan(arena, pairs)
        plan.run(mode, GO, GE)
        res[mode] = plan.scores()
        plan.close()
        assert np.isfinite(res[mode]).all()
    assert (res["semiglobal_both"] >= res["global"]).all()
    spot = np.sort(rng.choice(len(pairs), 24, replace=False))
    cat = np.concatenate(profs, axis=0)
    row_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    sc_or, _ = orc.batch_align(modes, cat, row_off, lens, S, pairs[spot], GO, GE, threads=min(8, host_threads()))
    for q, mode in enumerate(modes):
        assert np.array_equal(res[mode][spot], sc_or[:, q]), mode
    sub = rng.integers(0, len(pairs), 600)
    plan = nat.Plan(arena, pairs[sub][:, ::-1].copy())
    plan.run("global", GO, GE)
    assert np.array_equal(plan.scores(), res["global"][sub])
    plan.close()
    arena.close()


def packed_rows(buf, off, rows):
    """The rows of a packed path buffer that hold paths (the slack of every pair's slot is never written: it shows
    whatever the recycled device block held before)."""
    rows = np.asarray(rows, dtype=np.int64)
    start = np.cumsum(rows) - rows
    idx = np.repeat(np.asarray(off, dtype=np.int64) - start, rows) + np.arange(int(rows.sum()), dtype=np.int64)
    return buf[idx]


def test_chunked_path_plans_on_two_streams(nat, monkeypatch):
    """44 787 global alignments with paths of ~1 000 x 1 300 nucleotides (a batch `scripts/stress.py` drew): the scratch of
    such a plan - tens of GB - runs in chunks that alternate between two streams and two scratch sets.  Regression test:
    the sets used to be re-allocated inside the chunk loop, handing a block that an earlier chunk's kernels were still
    reading back to the pool - and on to the other stream's set (thousands of wrong paths, different from run to run).
    Single pass and two-pass scheme, default and smaller budgets, must give identical scores and paths, and sampled pairs
    must equal the oracle."""
    from praline_amd.matrices import nucleotide_matrix
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chunked_paths_dna.npz"))
    lens, pairs, sym = d["lens"], d["pairs"], d["sym"]
    off = np.concatenate([[0], np.cumsum(lens)])
    profs = [np.eye(15, dtype=np.float32)[sym[off[i]:off[i + 1]]] for i in range(len(lens))]
    S = nucleotide_matrix()
    arena = nat.Arena(profs, S)

    def run(two, budget, quad="0", pk16="0"):
        monkeypatch.setenv("PRALINE_TB_TWOPASS", two)
        monkeypatch.setenv("PRALINE_TB_QUAD", quad)    # "0": the strip kernels (single pass / two passes); "1": k_dp_quad_tb
        monkeypatch.setenv("PRALINE_TB_PK16", pk16)    # "1" (with quad "1"): k_dp_pk16_tb - (2 302 + 2) * 11 = 25 344 < 32 000
        if budget:
            monkeypatch.setenv("PRALINE_TB_BUDGET_MB", budget)
        else:
            monkeypatch.delenv("PRALINE_TB_BUDGET_MB", raising=False)
        plan = nat.Plan(arena, pairs, want_paths=True)
        plan.run("global", *GAPS)
        sc = plan.scores().copy()
        buf, o, r = plan.paths_packed()
        assert plan.kernel_name().startswith("k_dp_pk16_tb") == (quad == "1" and pk16 == "1"), plan.kernel_name()
        res = (sc, packed_rows(buf, o, r), o.copy(), r.copy(), plan.match_kind(), buf)
        plan.close()
        return res

    ref = run("0", "160000")                       # one chunk
    for two, budget, quad, pk16 in (("0", None, "0", "0"), ("2", None, "0", "0"), ("2", "3000", "0", "0"), ("0", "3000", "0", "0"), ("2", None, "0", "0"),
                                    ("0", None, "1", "0"), ("0", "3000", "1", "0"), ("0", "160000", "1", "0"),   # ... and the plans' default kernel
                                    ("0", None, "1", "1"), ("0", "3000", "1", "1")):                            # ... and packed int16
        res = run(two, budget, quad, pk16)
        assert np.array_equal(bits(res[0]), bits(ref[0])), (two, budget, quad, pk16)
        assert np.array_equal(res[3], ref[3]) and np.array_equal(res[2], ref[2]) and np.array_equal(res[1], ref[1]), (two, budget, quad, pk16)
    rng = np.random.default_rng(7)
    for k in rng.permutation(len(pairs))[:6]:
        i, j = pairs[k]
        s_or, p_or = oracle_dp_on_m("global", arena.match_scores(int(i), int(j), ref[4]))
        assert ref[0][k] == np.float32(s_or)
        assert np.array_equal(ref[5][ref[2][k]:ref[2][k] + ref[3][k]], p_or)
    arena.close()


def test_chain_mode_chunk_by_chunk(nat, bba, monkeypatch):
    """Path plans whose packed traceback exceeds the scratch budget run in chain mode (one wave per task and strip) chunk
    by chunk - long sequences: a chunk holds few tasks, which task mode would run one wave each.  Float profiles of
    ~600 aa under a 48 MB budget (three or four tasks per chunk), three modes: scores and every path equal task mode
    (PRALINE_NO_CHAIN=1) and the one-chunk run; sampled pairs equal the oracle."""
    from conftest import synth_profile, synth_lengths
    rng = np.random.default_rng(19)
    lens = synth_lengths(rng, 40, 600).astype(np.int32)
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    pairs = np.array([(i, j) for i in range(40) for j in range(40) if i != j], dtype=np.int32)
    arena = nat.Arena(profs, bba["S"])

    def run(mode, budget, no_chain):
        monkeypatch.setenv("PRALINE_TB_TWOPASS", "0")
        monkeypatch.setenv("PRALINE_TB_BUDGET_MB", budget)
        monkeypatch.setenv("PRALINE_NO_CHAIN", "1" if no_chain else "0")
        plan = nat.Plan(arena, pairs, want_paths=True)
        plan.run(mode, *GAPS)
        sc = plan.scores().copy()
        buf, o, r = plan.paths_packed()
        res = (sc, packed_rows(buf, o, r), o.copy(), r.copy(), plan.match_kind(), buf, plan.kernel_name())
        plan.close()
        return res

    for mode in ("global", "local", "semiglobal_one"):
        one = run(mode, "64000", False)
        chunks = run(mode, "48", False)
        tasks = run(mode, "48", True)
        for other in (chunks, tasks):
            assert np.array_equal(bits(other[0]), bits(one[0])), mode
            assert np.array_equal(other[3], one[3]) and np.array_equal(other[1], one[1]), mode
        for k in np.random.default_rng(3).permutation(len(pairs))[:4]:
            i, j = pairs[k]
            s_or, p_or = oracle_dp_on_m(mode, arena.match_scores(int(i), int(j), one[4]))
            assert chunks[0][k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(chunks[5][chunks[2][k]:chunks[2][k] + chunks[3][k]], p_or), (mode, i, j)
    arena.close()


def test_scratch_growth_between_back_to_back_path_runs(nat, monkeypatch):
    """The device-buffer pool is stream-ordered (csrc/praline_dp.hip, pool_release / pool_alloc): a scratch block that a
    run replaces while an earlier run's kernels - on either of the library's two streams - may still be using it is not
    handed to anyone before both streams have passed the release point.  Two plans, runs queued back to back WITHOUT any
    host synchronisation under budgets that make every run re-cut its chunks and GROW its scratch sets (small budget: many
    chunks on two streams; larger budget: fewer, larger blocks; a second plan picking blocks up in between): scores and
    every path of every run must equal the one-chunk reference."""
    from praline_amd.matrices import nucleotide_matrix
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chunked_paths_dna.npz"))
    lens, pairs, sym = d["lens"], d["pairs"], d["sym"]
    off = np.concatenate([[0], np.cumsum(lens)])
    profs = [np.eye(15, dtype=np.float32)[sym[off[i]:off[i + 1]]] for i in range(len(lens))]
    arena = nat.Arena(profs, nucleotide_matrix())
    half = len(pairs) // 2
    subsets = (pairs[:half], pairs[half // 2:half // 2 + half])

    def results(plan):
        sc = plan.scores().copy()
        buf, o, r = plan.paths_packed()
        return sc, packed_rows(buf, o, r), o.copy(), r.copy()

    monkeypatch.setenv("PRALINE_TB_BUDGET_MB", "160000")
    monkeypatch.setenv("PRALINE_TB_QUAD", "0")     # (the subject is the strip kernels' two scratch layouts)
    ref = []
    for sub in subsets:
        for two in ("0", "2"):
            monkeypatch.setenv("PRALINE_TB_TWOPASS", two)
            plan = nat.Plan(arena, sub, want_paths=True)
            plan.run("global", *GAPS)
            if two == "0":
                ref.append(results(plan))
            else:
                got = results(plan)
                assert all(np.array_equal(a, b) for a, b in zip(got, ref[-1]))
            plan.close()
    nat.pool_trim()
    for two in ("0", "2"):
        monkeypatch.setenv("PRALINE_TB_TWOPASS", two)
        plans = [nat.Plan(arena, sub, want_paths=True) for sub in subsets]
        outs = []
        # every run below is queued behind the previous one with no host wait in between; the budgets rise, so every
        # run replaces its scratch blocks with larger ones while the previous run is still executing
        for budget, which in (("1500", 0), ("2500", 1), ("4000", 0), ("9000", 1), ("20000", 0), ("160000", 1)):
            monkeypatch.setenv("PRALINE_TB_BUDGET_MB", budget)
            plans[which].run("global", *GAPS)
            if budget in ("9000", "20000", "160000"):
                outs.append((which, budget, results(plans[which])))   # (reading the results waits for that run only)
        for which, budget, got in outs:
            assert np.array_equal(bits(got[0]), bits(ref[which][0])), (two, which, budget)
            assert all(np.array_equal(a, b) for a, b in zip(got[1:], ref[which][1:])), (two, which, budget)
        for p in plans:
            p.close()
    arena.close()


def test_single_long_alignments(nat, bba):
    """The long end of the size range: ONE alignment of 6 211 x 9 001 nucleotides (55.9 M cells; lengths that are no
    multiple of the 32-column strips or the 12-row loop iterations), scores-only and with paths in all five modes, and
    one of 2 345 x 3 111 float profiles - plans of a single task, which the library runs with one wave per strip
    (chain mode; the scores-only form without flags).  Scores and paths equal the oracle's, which walks the whole
    matrices on the host."""
    d = load_golden("synthetic_dna.npz")
    Sd = d["matrix"]
    rng = np.random.default_rng(91)
    # integer scoring: the oracle computes match scores and alignment from the sequences alone
    v1, v2 = rng.integers(0, 4, 6211), rng.integers(0, 4, 9001)
    v2[1000:5000] = v1[800:4800]                      # a long common stretch: paths with long diagonals and real gaps
    profs = [one_hot(v1, 15), one_hot(v2, 15)]
    lens = np.array([len(v1), len(v2)], dtype=np.int32)
    arena = nat.Arena(profs, Sd)
    pair = np.array([[0, 1]], dtype=np.int32)
    cat = np.concatenate(profs, axis=0)
    row_off = np.array([0, len(v1)], dtype=np.int64)
    sc_or, paths_or = orc.batch_align(MODES5, cat, row_off, lens, Sd, pair, GO, GE, threads=min(5, host_threads()))
    p_scores = nat.Plan(arena, pair)
    p_paths = nat.Plan(arena, pair, want_paths=True)
    for q, mode in enumerate(MODES5):
        p_scores.run(mode, GO, GE)
        assert p_scores.scores()[0] == sc_or[0, q], (mode, "scores only")
        p_paths.run(mode, GO, GE)
        assert p_paths.scores()[0] == sc_or[0, q], (mode, "with paths")
        assert np.array_equal(p_paths.paths()[0], paths_or[0][q]), mode
    p_scores.close(); p_paths.close(); arena.close()
    # float profiles: the oracle's DP on the device's own match scores
    f1, f2 = synth_profile(rng, 2345)[0], synth_profile(rng, 3111)[0]
    arena = nat.Arena([f1, f2], bba["S"])
    p_scores = nat.Plan(arena, pair)
    p_paths = nat.Plan(arena, pair, want_paths=True)
    m = arena.match_scores(0, 1, p_paths.match_kind())
    for mode in MODES5:
        s_or, path_or = oracle_dp_on_m(mode, m)
        p_scores.run(mode, GO, GE)
        p_paths.run(mode, GO, GE)
        assert p_scores.scores()[0] == np.float32(s_or) and p_paths.scores()[0] == np.float32(s_or), mode
        assert np.array_equal(p_paths.paths()[0], path_or), mode
    p_scores.close(); p_paths.close(); arena.close()
