import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths
nat.init(0)
S = blosum62_matrix(); N = 512
rng = np.random.default_rng(3); lens = synth_lengths(rng, N, 250)
profs = []
for L in lens:
    p = np.zeros((L, 27), np.float32); p[np.arange(L), rng.integers(0, 20, L)] = 1; profs.append(p)
pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
ar = nat.Arena(profs, S)
rects = [[(20, 120, 30, 140)] for _ in pairs]
for name, kw in (("local", {}), ("local+mask", {"rects": rects}), ("global", {})):
    pl = nat.Plan(ar, pairs, want_paths=True, **kw)
    mode = "global" if name == "global" else "local"
    for rep in range(3):
        t1 = time.perf_counter(); pl.run(mode, -11, -1); nat.synchronize(); t2 = time.perf_counter()
    print("%-12s kernels %.1f ms (%.0f GCUPS)" % (name, (t2 - t1) * 1e3, cells / (t2 - t1) / 1e9), flush=True)
    pl.close()
