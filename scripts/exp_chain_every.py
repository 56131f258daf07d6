"""C2 float profiles with paths (chain mode): run time against the publish interval (PRALINE_CHAIN_EVERY), and the
task-mode / two-pass alternatives."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from bench import make_workload
nat.init(0)
w = make_workload("c2")
pairs = allpairs.enumerate_pairs(256)
cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
ar = nat.Arena(w["profs"], w["S"])
def run(label, env, mode="global"):
    for k in ("PRALINE_CHAIN_EVERY", "PRALINE_NO_CHAIN", "PRALINE_TB_TWOPASS", "PRALINE_TB_KEEP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    pl = nat.Plan(ar, pairs, want_paths=True)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): pl.run(mode, -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-28s %-6s %.2f ms  %.0f GCUPS  [%s]" % (label, mode, dt * 1e3, cells / dt / 1e9, pl.kernel_name()), flush=True)
    pl.close()
for ev in (12, 24, 48, 96, 144, 192, 400):
    run("chain every %d" % ev, {"PRALINE_CHAIN_EVERY": str(ev)})
run("task mode", {"PRALINE_NO_CHAIN": "1"})
run("two-pass", {"PRALINE_TB_TWOPASS": "2"})
run("keep forward", {"PRALINE_TB_KEEP": "1"})
run("chain (default)", {}, "local")
run("two-pass", {"PRALINE_TB_TWOPASS": "2"}, "local")
