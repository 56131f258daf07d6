"""A/B of library builds on alignments WITH paths: C2 float (chain mode), C2 one-hot, a C3 slice (task mode), three
modes; each build in its own process.  usage: exp_paths_ab.py default variants/libpraline_dp_x.so ..."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from praline_amd import native as nat
import bench
nat.init(0)
def timed(plan, mode):
    ts = []
    for _ in range(7):
        plan.run(mode, -11.0, -1.0); nat.synchronize(); ts.append(plan.kernel_ms())
    return float(np.median(ts[2:]))
out = []
w = bench.make_workload("c2")
n = len(w["lens"])
pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
cells = float((np.asarray(w["lens"])[pairs[:, 0]].astype(np.int64) * np.asarray(w["lens"])[pairs[:, 1]]).sum())
arena = nat.Arena(w["profs"], w["S"])
plan = nat.Plan(arena, pairs, want_paths=True)
for mode in ("global", "local"):
    ms = timed(plan, mode); out.append("c2f %%s %%.2f ms %%.0f" %% (mode[:3], ms, cells / ms / 1e6))
chk = float(plan.scores().astype(np.float64).sum())
plan.close(); arena.close()
rng = np.random.default_rng(5)
oh = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in w["lens"]]
arena = nat.Arena(oh, w["S"])
plan = nat.Plan(arena, pairs, want_paths=True)
ms = timed(plan, "global"); out.append("c2oh glo %%.2f ms %%.0f" %% (ms, cells / ms / 1e6))
plan.close(); arena.close()
# a C3-like slice: 1024 one-hot sequences ~250 aa, 131072 ordered pairs
rng = np.random.default_rng(6)
lens = bench.synth_lengths(rng, 1024, 250)
oh = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
arena = nat.Arena(oh, w["S"])
pairs = np.array([(i, j) for j in range(128) for i in range(1024) if i != j], dtype=np.int32)
cells = float((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
plan = nat.Plan(arena, pairs, want_paths=True)
for mode in ("global", "local", "semiglobal_both"):
    ms = timed(plan, mode); out.append("c3s %%s %%.2f ms %%.0f" %% (mode[:3], ms, cells / ms / 1e6))
print(" | ".join(out), "| chk %%.3f" %% chk)
''' % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default": env["PRALINE_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("%-34s" % lib, r.stdout.strip() or r.stderr.strip()[-600:], flush=True)
