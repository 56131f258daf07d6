"""Shared-wave workgroups: wave counts chosen by the modelled busiest SIMD (PRALINE_WG_BALANCE=1, default) against the
longest-wave threshold (=0): kernel time of the C2 launch, scores bitwise equal."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from bench import make_workload
nat.init(0)
for wl in sys.argv[1:] or ["c2"]:
    w = make_workload(wl)
    arena = nat.Arena(w["profs"], w["S"])
    n = len(w["lens"])
    pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
    if wl != "c2":
        pairs = pairs[pairs[:, 1] % 8 == 3]
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    ref = None
    for rep in range(2):
        for bal in ("0", "1"):
            os.environ["PRALINE_WG_BALANCE"] = bal
            t0 = time.perf_counter()
            plan = nat.Plan(arena, pairs, want_paths=False)
            t_plan = time.perf_counter() - t0
            for mode in ("global", "local"):
                ks = []
                for _ in range(12):
                    plan.run(mode, -11.0, -1.0)
                    nat.synchronize()
                    ks.append(plan.kernel_ms())
                sc = plan.scores().copy()
                key = (mode,)
                if ref is None: ref = {}
                if key in ref: assert np.array_equal(ref[key].view(np.uint32), sc.view(np.uint32)), "scores differ"
                ref[key] = sc
                k = float(np.median(ks[2:]))
                print("%s balance=%s %-6s plan %.2f ms  kernel %.3f ms  %.0f GCUPS  %s" % (wl, bal, mode, t_plan * 1e3, k, cells / k / 1e6, plan.kernel_name()), flush=True)
            plan.close()
