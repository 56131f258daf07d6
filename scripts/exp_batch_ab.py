import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/praline_amd") else os.getcwd())
from praline_amd import native as nat, allpairs
from bench import make_workload
nat.init(0)
w = make_workload("c2")
pairs = allpairs.enumerate_pairs(256)[::4]
cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
ar = nat.Arena(w["profs"], w["S"])
for paths in (False, True):
    for mode in ("global", "local"):
        pl = nat.Plan(ar, pairs, want_paths=paths)
        pl.run(mode, -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): pl.run(mode, -11, -1)
        nat.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("paths=%d %-6s %.2f ms %5.0f GCUPS [%s] checksum %.3f" % (paths, mode, dt * 1e3, cells / dt / 1e9, pl.kernel_name(), float(pl.scores().astype(np.float64).sum())), flush=True)
        pl.close()
