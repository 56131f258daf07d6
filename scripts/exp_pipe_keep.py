"""Two-pass alignments with paths whose forward fill runs on the PIPELINE kernel (k_dp_pipe<..., KEEP> + k_trace_recompute on
blocks of PRALINE_KEEP_BH rows) against the single pass (chain / task mode): identical scores and paths on random
float-profile batches (global mode), then the C2 rate with the kernel split."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PRALINE_PIPE_MIN_TASKS", "1")
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from conftest import synth_profile
from bench import make_workload
nat.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)

def run(arena, pairs, mode, pipe):
    os.environ["PRALINE_TB_PIPE"] = "1" if pipe else "0"
    plan = nat.Plan(arena, pairs, want_paths=True)
    plan.run(mode, -11.0, -1.0)
    sc = plan.scores().copy()
    buf, off, rows = plan.paths_packed()
    name = plan.kernel_name()
    plan.close()
    return sc, buf, off, rows, name

n_cases = n_pipe = 0
t_end = time.time() + float(os.environ.get("SECONDS", "60"))
t_print = time.time()
while time.time() < t_end:
    N = int(rng.choice([2, 5, 17, 40, 90]))
    mu = int(rng.choice([3, 20, 40, 70, 130, 260, 520]))
    lens = np.maximum(1, rng.integers(max(1, mu // 2), mu * 3 // 2 + 1, N))
    S = blosum62_matrix(); profs = [synth_profile(rng, int(L))[0] for L in lens]
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < rng.choice([0.3, 1.0])]
    if len(pairs) == 0:
        continue
    arena = nat.Arena(profs, S)
    a = run(arena, pairs, "global", False)
    b = run(arena, pairs, "global", True)
    arena.close()
    n_cases += 1
    if not b[4].startswith("k_dp_pipe"):
        continue   # (the pair list did not suit the pipeline layout)
    n_pipe += 1
    if not np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)):
        k = int(np.flatnonzero(a[0] != b[0])[0])
        print("SCORE MISMATCH", N, mu, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], a[0][k], b[0][k]); sys.exit(1)
    for k in range(len(pairs)):
        pa = a[1][a[2][k]:a[2][k] + a[3][k]]; pb = b[1][b[2][k]:b[2][k] + b[3][k]]
        if not np.array_equal(pa, pb):
            print("PATH MISMATCH", N, mu, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], len(pa), len(pb))
            d = min(len(pa), len(pb))
            bad = [q for q in range(1, d + 1) if not np.array_equal(pa[-q], pb[-q])]
            print(" first difference from the end at", bad[:1], pa[-(bad[0] if bad else 1)], pb[-(bad[0] if bad else 1)])
            sys.exit(1)
    if time.time() - t_print > 30:
        print("  ...", n_cases, "batches,", n_pipe, "on the pipeline", flush=True); t_print = time.time()
print("pipeline two-pass == single pass on %d random batches (%d of %d took the pipeline)" % (n_pipe, n_pipe, n_cases), flush=True)
if os.environ.get("RATES", "1") == "1":
    os.environ.pop("PRALINE_PIPE_MIN_TASKS", None)
    w = make_workload("c2")
    arena = nat.Arena(w["profs"], w["S"])
    n = len(w["lens"])
    pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    res = {}
    for pipe in ("0", "1"):
        os.environ["PRALINE_TB_PIPE"] = pipe
        plan = nat.Plan(arena, pairs, want_paths=True)
        for _ in range(2):
            plan.run("global", -11.0, -1.0)
        nat.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            plan.run("global", -11.0, -1.0)
        nat.synchronize()
        dt = (time.perf_counter() - t) / 5
        print("C2 global with paths PIPE=%s  %s  %.3f ms  %.0f GCUPS" % (pipe, plan.kernel_name(), dt * 1e3, cells / dt / 1e9), flush=True)
        res[pipe] = (plan.scores().copy(), plan.paths_packed())
        plan.close()
    assert np.array_equal(res["0"][0].view(np.uint32), res["1"][0].view(np.uint32)), "C2 scores differ"
    for q in range(3):
        assert np.array_equal(res["0"][1][q], res["1"][1][q]), "C2 paths differ (%d)" % q
    print("C2: scores and paths identical", flush=True)
