import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
for N, kind in ((128, "onehot"), (256, "onehot"), (256, "profile")):
    rng = np.random.default_rng(3)
    lens = synth_lengths(rng, N, 250)
    profs = []
    for L in lens:
        if kind == "onehot":
            p = np.zeros((L, 27), np.float32); p[np.arange(L), rng.integers(0, 20, L)] = 1
        else:
            p = synth_profile(rng, int(L))
        profs.append(p)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S)
    for mode in ("global", "local", "semiglobal_both"):
        pl = nat.Plan(ar, pairs, want_paths=True)
        pl.run(mode, -11, -1); nat.synchronize()
        t0 = time.perf_counter(); pl.run(mode, -11, -1); nat.synchronize(); t1 = time.perf_counter()
        print("N=%d %-8s %-16s pairs=%d cells=%.3g  %.2f ms  %.0f GCUPS (fill + traceback kernels)" % (N, kind, mode, len(pairs), cells, (t1-t0)*1e3, cells/(t1-t0)/1e9))
        pl.close()
    ar.close()
