"""GPU tests at BASELINE.json's full configuration sizes (C3, one rank's share of C4, C5): properties that
do not need the oracle at full size - the score recomputed from the returned path, mode ordering, symmetry
and self-alignment under integer scoring - plus oracle spot checks on sampled pairs."""
import numpy as np
import pytest

from conftest import load_golden, one_hot, synth_lengths, synth_profile
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GO, GE = -11.0, -1.0


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


def test_c3_preprofile_modes_full_size(nat, bba):
    """BASELINE config 2: 1024 seqs ~250 aa, ordered pairs, local and semiglobal with paths (one pass over
    a 200k-pair slice per mode keeps the test short; the full list runs in scripts/exp_c3.py)."""
    rng = np.random.default_rng(3)
    N = 1024
    lens = synth_lengths(rng, N, 250)
    vals = [rng.integers(0, 20, int(L)) for L in lens]
    S = bba["S"]
    arena = nat.Arena([one_hot(v, 27) for v in vals], S)
    allp = np.array([(i, j) for i in range(0, N, 5) for j in range(N) if i != j], dtype=np.int32)   # 209 k ordered pairs
    results = {}
    for mode in ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"):
        plan = nat.Plan(arena, allp, want_paths=True)
        plan.run(mode, GO, GE)
        sc = plan.scores()
        buf, off, rows = plan.paths_packed()
        plan.close()
        results[mode] = sc
        assert np.isfinite(sc).all()
        for k in rng.integers(0, len(allp), 60):
            i, j = allp[k]
            path = buf[off[k]:off[k] + rows[k]]
            if mode == "global" or mode.startswith("semiglobal"):
                assert tuple(path[0]) == (0, 0) and tuple(path[-1]) == (lens[i], lens[j]), (mode, i, j)
            assert path_score(path, vals[i], vals[j], S, mode) == sc[k], (mode, i, j)
        for k in rng.integers(0, len(allp), 6):
            i, j = allp[k]
            s_or, p_or = orc.pairwise_score_fast(mode, one_hot(vals[i], 27), one_hot(vals[j], 27), S, GO, GE, want_path=True)
            assert sc[k] == np.float32(s_or), (mode, i, j)
            assert np.array_equal(buf[off[k]:off[k] + rows[k]], p_or), (mode, i, j)
    # (local >= semiglobal_both does NOT hold in the reference: local mode keeps the penalised global boundary,
    # align.py:371-385, so an alignment that starts in the first row or column scores less than with free ends)
    for hi, lo in (("semiglobal_both", "semiglobal_one"), ("semiglobal_both", "semiglobal_two"),
                   ("semiglobal_one", "global"), ("semiglobal_two", "global")):
        bad = np.nonzero(~(results[hi] >= results[lo]))[0]
        assert len(bad) == 0, (hi, lo, len(bad), allp[bad[:3]].tolist(), results[hi][bad[:3]], results[lo][bad[:3]])
    arena.close()


def test_c4_rank_share_full_size(nat, bba):
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
    # float profiles on a 150 k-pair part of the same shard
    profs = [synth_profile(rng, int(L))[0] for L in lens]
    arena = nat.Arena(profs, S)
    part = mine[::7]
    plan = nat.Plan(arena, part)
    plan.run("global", GO, GE)
    scf = plan.scores()
    plan.close()
    assert np.isfinite(scf).all()
    for k in rng.integers(0, len(part), 4):
        i, j = part[k]
        ref = orc.pairwise_score_fast("global", profs[i], profs[j], S, GO, GE)
        assert abs(scf[k] - ref) <= 1e-5 * abs(ref)
    arena.close()


def test_c5_long_dna_full_size(nat):
    """BASELINE config 4: 512 nucleotide seqs ~5 kb (3.3e12 cells for all pairs): a 9 k-pair column shard
    (1/14 of the list, 2.3e11 cells) in global mode, exact against the oracle on sampled pairs; symmetry."""
    from praline_amd import allpairs
    d = load_golden("synthetic_dna.npz")
    S = d["matrix"]
    rng = np.random.default_rng(5)
    N = 512
    lens = synth_lengths(rng, N, 5000)
    vals = [rng.integers(0, 4, int(L)) for L in lens]
    arena = nat.Arena([one_hot(v, 15) for v in vals], S)
    pairs = allpairs.enumerate_pairs(N)
    mine = pairs[allpairs.shard_columns(lens, pairs, 14)[5]]
    plan = nat.Plan(arena, mine)
    plan.run("global", GO, GE)
    sc = plan.scores()
    plan.close()
    assert np.isfinite(sc).all()
    for k in rng.integers(0, len(mine), 3):
        i, j = mine[k]
        assert sc[k] == np.float32(orc.pairwise_score_fast("global", one_hot(vals[i], 15), one_hot(vals[j], 15), S, GO, GE))
    sub = rng.integers(0, len(mine), 300)
    plan = nat.Plan(arena, mine[sub][:, ::-1].copy())
    plan.run("global", GO, GE)
    assert np.array_equal(plan.scores(), sc[sub])
    plan.close()
    arena.close()
