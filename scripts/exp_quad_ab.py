"""A/B of library variants (PRALINE_LIB) on a C3 slice, one-hot, alignments with paths on k_dp_quad_tb: kernel time per mode
(HIP events around the run: fill + traceback).  usage: python scripts/exp_quad_ab.py [variant.so ...]"""
import sys, os, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("QA_CHILD") == "1":
    sys.path.insert(0, ROOT)
    import numpy as np
    from praline_amd import native as nat
    from bench import make_workload, one_hot, synth_lengths
    nat.init(0)
    S = make_workload("c2")["S"]
    rng3 = np.random.default_rng(3)
    l3 = synth_lengths(rng3, 1024, 250)
    a3 = nat.Arena([one_hot(rng3.integers(0, 20, int(L)), 27) for L in l3], S)
    i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
    p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
    frac = int(os.environ.get("C3_FRAC", "4"))
    p3 = p3[p3[:, 1] % frac == 1]
    c3 = int((l3[p3[:, 0]].astype(np.int64) * l3[p3[:, 1]]).sum())
    out = []
    for mode in os.environ.get("MODES", "global,local,semiglobal_both").split(","):
        plan = nat.Plan(a3, p3, want_paths=True)
        for _ in range(2):
            plan.run(mode, -11.0, -1.0)
        nat.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); plan.run(mode, -11.0, -1.0); nat.synchronize(); best = min(best, time.perf_counter() - t)
        out.append("%s %.2f ms %.0f" % (mode[:6], best * 1e3, c3 / best / 1e9))
        name = plan.kernel_name()
        plan.close()
    print("%-34s %-22s %s" % (os.environ.get("PRALINE_LIB", "default")[-34:], name, " | ".join(out)), flush=True)
    sys.exit(0)
libs = sys.argv[1:] or [""]
for rep in range(int(os.environ.get("REPS", "2"))):
    for lib in libs:
        env = dict(os.environ, QA_CHILD="1")
        if lib:
            env["PRALINE_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
