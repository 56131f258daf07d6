"""Members (dz, place) of the balanced share family (sched.cpp) on C2: modelled makespan against measured kernel time."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from bench import make_workload
nat.init(0)
w = make_workload("c2")
arena = nat.Arena(w["profs"], w["S"])
n = len(w["lens"])
pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
def measure(tag):
    plan = nat.Plan(arena, pairs, want_paths=False)
    ks = []
    for _ in range(14):
        plan.run("global", -11.0, -1.0); nat.synchronize(); ks.append(plan.kernel_ms())
    plan.close()
    print("%-14s kernel %.3f ms (min %.3f)" % (tag, float(np.median(ks[2:])), min(ks)), flush=True)
os.environ["PRALINE_WG_BALANCE"] = "0"
measure("threshold")
os.environ["PRALINE_WG_BALANCE"] = "1"
os.environ["PRALINE_SCHED_DEBUG"] = "1"
for dz in (-12, -8, -6, -4, -2, 0):
    for place in (0, 2, 4, 6, 10, 14):
        os.environ["PRALINE_WG_BALANCE_FORCE"] = "%d,%d" % (dz, place)
        measure("dz=%d p=%d" % (dz, place))
os.environ["PRALINE_WG_BALANCE"] = "0"
measure("threshold")
