"""C2: the flag-free chain fill (PRALINE_SCORES_CHAIN=1) as a candidate forward pass of a two-pass path scheme."""
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
for env, paths in (({"PRALINE_SCORES_CHAIN": "1", "PRALINE_NO_PIPE": "1"}, False), ({"PRALINE_SCORES_CHAIN": "0", "PRALINE_NO_PIPE": "1"}, False), ({}, False), ({}, True)):
    for k in ("PRALINE_SCORES_CHAIN", "PRALINE_NO_PIPE"): os.environ.pop(k, None)
    os.environ.update(env)
    for mode in ("global", "local"):
        pl = nat.Plan(ar, pairs, want_paths=paths)
        pl.run(mode, -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): pl.run(mode, -11, -1)
        nat.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-60s paths=%d %-6s %.2f ms %5.0f GCUPS [%s]" % (env, paths, mode, dt * 1e3, cells / dt / 1e9, pl.kernel_name()), flush=True)
        pl.close()
