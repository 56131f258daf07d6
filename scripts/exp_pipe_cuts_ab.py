"""A/B on one box: an older build of the library (variants/libpraline_dp_head.so) against the current one with whole
tasks (PRALINE_PIPE_CUTS=0) and with cut tasks, C2 in three modes.  Each build in its own process."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from praline_amd import native as nat
import bench
nat.init(0)
w = bench.make_workload("c2")
arena = nat.Arena(w["profs"], w["S"])
n = len(w["lens"])
pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
plan = nat.Plan(arena, pairs)
out = []
for mode in ("global", "local", "semiglobal_both"):
    ks = []
    for _ in range(40):
        plan.run(mode, -11.0, -1.0); nat.synchronize(); ks.append(plan.kernel_ms())
    out.append("%%s %%.3f" %% (mode, float(np.median(ks[5:]))))
print(" | ".join(out), "| cuts", plan.cut_tasks, "| checksum %%.3f" %% float(plan.scores().astype(np.float64).sum()))
''' % ROOT
for rep in range(3):
    for tag, env_add in (("head", {"PRALINE_LIB": os.path.join(ROOT, "variants", "libpraline_dp_head.so")}),
                         ("new whole", {"PRALINE_PIPE_CUTS": "0"}), ("new cuts", {})):
        env = dict(os.environ); env.update(env_add)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("%-10s" % tag, r.stdout.strip() or r.stderr.strip()[-400:], flush=True)
