"""A/B of library variants on C2 float profiles, global alignments with paths (two-pass on the pipeline): total ms per run and
the forward / recompute split from HIP events is not available per kernel, so run this under scripts/prof_trace.sh for the split.
usage: python scripts/exp_pk_ab.py [variant.so ...]   (each in a child process, PRALINE_LIB)"""
import sys, os, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("PK_CHILD") == "1":
    sys.path.insert(0, ROOT)
    import numpy as np
    from praline_amd import native as nat
    from bench import make_workload
    nat.init(0)
    w = make_workload("c2")
    arena = nat.Arena(w["profs"], w["S"])
    n = len(w["lens"])
    pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    plan = nat.Plan(arena, pairs, want_paths=True)
    for _ in range(3):
        plan.run("global", -11.0, -1.0)
    nat.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(5):
            plan.run("global", -11.0, -1.0)
        nat.synchronize()
        best = min(best, (time.perf_counter() - t) / 5)
    sc = plan.scores()
    print("%-40s %-44s %.3f ms  %.0f GCUPS  chk %.3f" % (os.environ.get("PRALINE_LIB", "default")[-40:], plan.kernel_name(), best * 1e3, cells / best / 1e9, float(sc.astype(np.float64).sum())), flush=True)
    sys.exit(0)
libs = sys.argv[1:] or [""]
for rep in range(int(os.environ.get("REPS", "2"))):
    for lib in libs:
        env = dict(os.environ, PK_CHILD="1")
        if lib:
            env["PRALINE_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
