"""Host time of arena + plan creation against the kernel time for the large configurations (C4 rank share, all of C4,
a C3-sized path plan): what a caller pays around one submission."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
import bench
nat.init(0)
w = bench.make_workload("c4")
lens = np.asarray(w["lens"]); n = len(lens)
t0 = time.perf_counter(); arena = nat.Arena(w["profs"], w["S"]); t1 = time.perf_counter()
print("C4 arena (4096 profiles, %.0f MB): %.1f ms" % (sum(p.nbytes for p in w["profs"]) / 1e6, (t1 - t0) * 1e3), flush=True)
iu = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
for tag, pairs in (("C4 share 1/8 (columns)", iu[iu[:, 1] % 8 == 3]), ("C4 all pairs", iu)):
    cells = float((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    for rep in range(2):
        t0 = time.perf_counter(); plan = nat.Plan(arena, pairs); t1 = time.perf_counter()
        plan.run("global", -11.0, -1.0); nat.synchronize()
        plan.run("global", -11.0, -1.0); nat.synchronize(); k = plan.kernel_ms()
        t2 = time.perf_counter(); sc = plan.scores(); t3 = time.perf_counter()
        plan.close()
        print("%s: %d pairs, plan %.1f ms, kernel %.1f ms (%.0f GCUPS), scores D2H %.1f ms" % (tag, len(pairs), (t1 - t0) * 1e3, k, cells / k / 1e6, (t3 - t2) * 1e3), flush=True)
arena.close()
rng = np.random.default_rng(3)
lens = bench.synth_lengths(rng, 1024, 250)
oh = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
arena = nat.Arena(oh, w["S"])
pairs = np.array([(i, j) for j in range(1024) for i in range(1024) if i != j], dtype=np.int32)
cells = float((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
for rep in range(2):
    t0 = time.perf_counter(); plan = nat.Plan(arena, pairs, want_paths=True); t1 = time.perf_counter()
    plan.run("global", -11.0, -1.0); nat.synchronize()
    ta = time.perf_counter(); plan.run("global", -11.0, -1.0); nat.synchronize(); tb = time.perf_counter()
    plan.close()
    print("C3 with paths: %d pairs, plan %.1f ms, run %.1f ms (%.0f GCUPS)" % (len(pairs), (t1 - t0) * 1e3, (tb - ta) * 1e3, cells / (tb - ta) / 1e9), flush=True)
arena.close()
