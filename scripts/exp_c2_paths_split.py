"""C2 float profiles with paths: single pass (chain mode) against the forced two-pass scheme, for a kernel-trace
breakdown (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from bench import make_workload
nat.init(0)
w = make_workload("c2")
arena = nat.Arena(w["profs"], w["S"])
n = len(w["lens"])
iu = np.triu_indices(n, 1)
pairs = np.stack(iu, axis=1).astype(np.int32)
cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
for tp in sys.argv[1:] or ["0", "2"]:
    os.environ["PRALINE_TB_TWOPASS"] = tp
    plan = nat.Plan(arena, pairs, want_paths=True)
    for _ in range(2):
        plan.run("global", -11.0, -1.0)
    nat.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        plan.run("global", -11.0, -1.0)
    nat.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("TWOPASS=%s  %s  %.3f ms  %.0f GCUPS" % (tp, plan.kernel_name(), dt * 1e3, cells / dt / 1e9), flush=True)
    plan.close()
