"""Host-to-host time of one C2 submission (profiles on the host -> scores on the host), phases, best of 5."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native
import bench
native.init(0)
w = bench.make_workload("c2")
profs, S, lens = w["profs"], w["S"], np.asarray(w["lens"])
n = len(lens)
pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
reps = []
for _ in range(7):
    t_a = time.perf_counter()
    prep = native.prepare_schedule_async([len(p) for p in profs], pairs)
    ar = native.Arena(profs, S)
    t_b = time.perf_counter()
    pl = native.Plan(ar, pairs, prepared=prep)
    t_c = time.perf_counter()
    pl.run("global", -11.0, -1.0)
    sc = pl.scores()
    t_d = time.perf_counter()
    pl.close(); ar.close()
    reps.append((t_b - t_a, t_c - t_b, t_d - t_c))
    print("arena %.3f  plan %.3f  run+d2h %.3f  total %.3f ms" % tuple(1e3 * x for x in (reps[-1] + (sum(reps[-1]),))), flush=True)
