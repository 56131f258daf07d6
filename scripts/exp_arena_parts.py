"""Arena creation on C2 with the concatenation uploaded in 1 .. 4 parts (native.Arena._create_in_parts, PRALINE_ARENA_PARTS)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
import bench
nat.init(0)
w = bench.make_workload("c2")
profs, S = w["profs"], w["S"]
for parts in ("1", "2", "3", "4", "2", "4", "1", "2"):
    os.environ["PRALINE_ARENA_PARTS"] = parts
    if parts == "1":
        nat._PARTS_MIN_BYTES = 1 << 40
    else:
        nat._PARTS_MIN_BYTES = 4 << 20
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); a = nat.Arena(profs, S); t1 = time.perf_counter(); a.close(); ts.append(t1 - t0)
    print("parts", parts, "arena median %.3f ms  min %.3f" % (np.median(ts) * 1e3, min(ts) * 1e3), flush=True)
