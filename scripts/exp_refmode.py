"""Throughput of the reference-order match-score mode (PRALINE_MATCH_REFERENCE) on C2 (float profiles)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from bench import make_workload
nat.init(0)
w = make_workload("c2")
ii, jj = np.triu_indices(256, k=1)
pairs = np.stack([ii, jj], axis=1).astype(np.int32)
cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
nat.set_match_mode("ref")
ar = nat.Arena(w["profs"], w["S"])
for paths in (False, True):
    pl = nat.Plan(ar, pairs, want_paths=paths)
    pl.run("global", -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): pl.run("global", -11, -1)
    nat.synchronize(); t1 = time.perf_counter()
    print("C2 ref mode paths=%d: %.1f ms  %.1f GCUPS" % (paths, (t1 - t0) / 2 * 1e3, cells * 2 / (t1 - t0) / 1e9), flush=True)
    pl.close()
