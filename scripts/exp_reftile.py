"""Reference-order plans on the split-strip kernels (k_match_tile + dense-tile instances) against the one-cell-per-thread
kernels' tiles (PRALINE_NO_REFTILE=1): bitwise equality of scores and paths, then the rates on C2."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from bench import make_workload
MODES = ["global", "semiglobal_both", "semiglobal_one", "semiglobal_two", "local"]
nat.init(0)
nat.set_match_mode("ref")


def run(ar, pairs, mode, paths, tile, rects=None):
    os.environ["PRALINE_NO_REFTILE"] = "0" if tile else "1"
    pl = nat.Plan(ar, pairs, want_paths=paths, rects=rects)
    pl.run(mode, -11, -1)
    nat.synchronize()
    out = (pl.scores().copy(), [p.copy() for p in pl.paths()] if paths else None, pl.kernel_name())
    pl.close()
    return out


def check(name, profs, S, pairs, set_sizes=None, rects=None):
    ar = nat.Arena(profs, S, set_sizes=set_sizes) if set_sizes else nat.Arena(profs, S)
    for mode in MODES:
        for paths in (False, True):
            if rects is not None and not paths:
                continue
            a = run(ar, pairs, mode, paths, True, rects)
            b = run(ar, pairs, mode, paths, False, rects)
            same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
            bad = 0 if not paths else sum(not np.array_equal(x, y) for x, y in zip(a[1], b[1]))
            print("%-10s %-16s paths=%d  scores %s  paths differing %d   [%s | %s]" % (name, mode, paths, "EQUAL" if same else "DIFFER (%d)" % int((a[0] != b[0]).sum()), bad, a[2], b[2]), flush=True)
    ar.close()


if len(sys.argv) < 2 or sys.argv[1] != "rate":
    from conftest import synth_profile, synth_lengths
    rng = np.random.default_rng(5)
    S = make_workload("c1")["S"] if False else None
    w = make_workload("c2")
    S = w["S"]
    lens = [41, 70, 33, 64, 9, 130, 257, 1, 2, 31, 32, 33, 127, 128, 129, 300]
    profs = [synth_profile(rng, L)[0] for L in lens]
    print("nonzeros per row max:", max(int((p != 0).sum(axis=1).max()) for p in profs), "A =", S.shape[0])
    pairs = np.array([(i, j) for i in range(len(lens)) for j in range(len(lens)) if i != j], dtype=np.int32)
    check("small", profs, S, pairs)
    rects = [[(3, 8, 2, 9)] if k % 2 else [] for k in range(len(pairs))]
    check("rects", profs, S, pairs, rects=rects)
    # two track sets
    p3 = []
    for L in lens:
        c = np.zeros((L, 3), dtype=np.float32)                      # one nonzero per row: 8 per row with the first set's 7
        c[np.arange(L), rng.integers(0, 3, L)] = rng.uniform(0.5, 1.5, L).astype(np.float32)
        p3.append(c)
    S3 = rng.normal(0, 2, (3, 3)).astype(np.float32)
    A = S.shape[0]
    S2 = np.zeros((A + 3, A + 3), dtype=np.float32)
    S2[:A, :A] = S
    S2[A:, A:] = S3
    cat = [np.concatenate([a, b], axis=1) for a, b in zip(profs, p3)]
    check("multiset", cat, S2, pairs, set_sizes=[A, 3])
    # C2 sample
    ii, jj = np.triu_indices(256, k=1)
    allp = np.stack([ii, jj], axis=1).astype(np.int32)
    sel = np.sort(np.random.default_rng(7).choice(len(allp), 4096, replace=False))
    check("c2-sample", w["profs"], S, allp[sel])
else:
    w = make_workload("c2")
    ii, jj = np.triu_indices(256, k=1)
    pairs = np.stack([ii, jj], axis=1).astype(np.int32)
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    ar = nat.Arena(w["profs"], w["S"])
    for paths in (False, True):
        for mode in ("global", "local"):
            pl = nat.Plan(ar, pairs, want_paths=paths)
            pl.run(mode, -11, -1); nat.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): pl.run(mode, -11, -1)
            nat.synchronize(); t1 = time.perf_counter()
            print("C2 ref mode %s paths=%d: %.1f ms  %.1f GCUPS  [%s]" % (mode, paths, (t1 - t0) / 3 * 1e3, cells * 3 / (t1 - t0) / 1e9, pl.kernel_name()), flush=True)
            pl.close()
