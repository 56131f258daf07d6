"""The alignments-with-paths kernels under a profiler: C2 float profiles (256 x ~400 aa, all 32 640 pairs, global: the pipeline
two-pass, k_dp_pipe<KEEP> + k_trace_recompute<36>; local: chain mode of k_dp_split16_tb), C2 one-hot (k_dp_pk16_tb in chain
mode) and a C3 slice (N one-hot sequences ~250 aa, all ordered pairs: k_dp_pk16_tb, one wave per task; k_dp_quad_tb beside it
with PRALINE_TB_PK16=0).  scripts/profile_paths.sh runs it under rocprofv3 --kernel-trace --stats and, in separate passes,
--pmc."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def run(tag, ar, pairs, lens, modes, reps=3):
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    for mode in modes:
        pl = nat.Plan(ar, pairs, want_paths=True)
        pl.run(mode, -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): pl.run(mode, -11, -1)
        nat.synchronize(); t1 = time.perf_counter()
        print("%s %-16s pairs=%d cells=%.3g  %.2f ms  %.0f GCUPS  [%s]" % (tag, mode, len(pairs), cells, (t1 - t0) / reps * 1e3, cells * reps / (t1 - t0) / 1e9, pl.kernel_name()), flush=True)
        pl.close()
rng = np.random.default_rng(2); lens = synth_lengths(rng, 256, 400)
ar = nat.Arena([synth_profile(rng, int(L)) for L in lens], S)
run("C2-float", ar, allpairs.enumerate_pairs(256), lens, ("global", "local"))
ar.close()
rng1 = np.random.default_rng(2)
ar = nat.Arena([np.eye(27, dtype=np.float32)[rng1.integers(0, 20, int(L))] for L in lens], S)
run("C2-onehot", ar, allpairs.enumerate_pairs(256), lens, ("global", "local"))
ar.close()
N = int(os.environ.get("N", "512"))
rng = np.random.default_rng(3); lens = synth_lengths(rng, N, 250)
ar = nat.Arena([np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens], S)
pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
run("C3-slice-N%d" % N, ar, pairs, lens, ("global", "local"))
os.environ["PRALINE_TB_PK16"] = "0"
run("C3-slice-N%d-quad" % N, ar, pairs, lens, ("global",))
os.environ.pop("PRALINE_TB_PK16")
ar.close()
