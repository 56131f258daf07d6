import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from oracle import oracle as orc
from conftest import one_hot
nat.init(0)
S = blosum62_matrix()
rng = np.random.default_rng(5)
lens = [60, 75, 48, 66, 90, 170, 159]
profs = [one_hot(rng.integers(0, 20, L), 27) for L in lens]
gaps = [np.stack([-rng.integers(8, 60, L) / 4.0, -rng.integers(1, 12, L) / 4.0], axis=1).astype(np.float32) for L in lens]
n = len(lens)
pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
for nrect_max in (3, 9):
    rects = []
    for k, (i, j) in enumerate(pairs):
        r = []
        for _ in range([0, 1, 3, nrect_max][k % 4]):
            y0, x0 = int(rng.integers(1, lens[i])), int(rng.integers(1, lens[j]))
            r.append((y0, min(lens[i], y0 + int(rng.integers(0, 12))), x0, min(lens[j], x0 + int(rng.integers(0, 12)))))
        rects.append(r)
    for use_gaps in (False, True):
        arena = nat.Arena(profs, S)
        if use_gaps: arena.set_gap_scores(gaps)
        for mode in ("local", "global"):
            plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
            mk = plan.match_kind()
            if use_gaps: plan.run_gaps(mode)
            else: plan.run(mode, -11.0, -1.0)
            sc, paths, kn = plan.scores(), plan.paths(), plan.kernel_name()
            plan.close()
            bad = 0
            for k, (i, j) in enumerate(pairs):
                zero = [(y, x) for (y0, y1, x0, x1) in rects[k] for y in range(y0, y1 + 1) for x in range(x0, x1 + 1)]
                g1, g2 = (gaps[i], gaps[j]) if use_gaps else orc.gap_arrays(lens[i], lens[j], (-11.0, -1.0))
                s_or, p_or = orc.raw_pairwise_align(mode, arena.match_scores(int(i), int(j), mk), g1, g2, zero or None)
                if sc[k] != np.float32(s_or) or not np.array_equal(paths[k], p_or):
                    bad += 1
                    if bad <= 3: print("   pair", (i, j), "rects", len(rects[k]), "dev", sc[k], "oracle", s_or)
            print("max rects %d gaps=%s %-6s [%s]: %d of %d differ" % (nrect_max, use_gaps, mode, kn, bad, len(pairs)), flush=True)
        arena.close()
