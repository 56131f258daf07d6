"""One C3-like slice with paths (1024 one-hot sequences ~250 aa, 131 072 ordered pairs, task mode), global mode: the
command behind the per-kernel profiles of the path kernels (rocprofv3 --kernel-trace --stats -- python3 scripts/exp_c3_slice.py)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
import bench
nat.init(0)
rng = np.random.default_rng(6)
lens = bench.synth_lengths(rng, 1024, 250)
oh = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
arena = nat.Arena(oh, blosum62_matrix())
pairs = np.array([(i, j) for j in range(128) for i in range(1024) if i != j], dtype=np.int32)
cells = float((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
plan = nat.Plan(arena, pairs, want_paths=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "global"
ts = []
for _ in range(6):
    plan.run(mode, -11.0, -1.0); nat.synchronize(); ts.append(plan.kernel_ms())
print("%s: %.2f ms, %.0f GCUPS" % (mode, float(np.median(ts[2:])), cells / float(np.median(ts[2:])) / 1e6))
