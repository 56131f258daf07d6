"""With-paths pipeline on a C3-like slice (N seqs ~250 aa one-hot, ordered pairs): run under rocprofv3
--kernel-trace --stats to see how the time splits between fill and traceback kernels."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths
nat.init(0)
S = blosum62_matrix()
N = int(os.environ.get("N", "384"))
rng = np.random.default_rng(3)
lens = synth_lengths(rng, N, 250)
profs = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
ar = nat.Arena(profs, S)
for mode in ("global", "local", "semiglobal_both"):
    pl = nat.Plan(ar, pairs, want_paths=True)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): pl.run(mode, -11, -1)
    nat.synchronize(); t1 = time.perf_counter()
    print("N=%d %-16s pairs=%d cells=%.3g  %.2f ms  %.0f GCUPS" % (N, mode, len(pairs), cells, (t1-t0)/3*1e3, cells*3/(t1-t0)/1e9), flush=True)
    pl.close()
ar.close()
