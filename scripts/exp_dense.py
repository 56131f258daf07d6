"""Plans whose fill reads dense match-score tiles (plan_run_dense) on BASELINE C2: per-position gap scores (fp32 MFMA tiles),
reference order (k_match_tile / one thread per cell) and a 40-symbol alphabet; scores-only and with paths.  Prints the rate of
each; run under scripts/prof_trace.sh for the split between the tile producer and the fill."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import make_workload
import torch
nat.init(0)
what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gaps", "ref", "cell", "wide"]
wl = make_workload("c2")
profs, S, lens = wl["profs"], wl["S"], np.asarray(wl["lens"])
n = len(lens)
pairs = np.array([(i, j) for i in range(n) for j in range(i + 1, n)], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
rng = np.random.default_rng(5)

def rate(plan, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return cells / ((time.perf_counter() - t) / reps) / 1e9

for w in what:
    if w == "gaps":
        arena = nat.Arena(profs, S)
        arena.set_gap_scores([np.stack([-rng.uniform(8.0, 14.0, int(L)), -rng.uniform(0.5, 2.0, int(L))], axis=1).astype(np.float32) for L in lens])
        run = lambda pl: pl.run_gaps("global")
    elif w in ("ref", "cell"):
        nat.set_match_mode("ref")
        if w == "cell":
            os.environ["PRALINE_NO_REFTILE"] = "1"
        arena = nat.Arena(profs, S)
        run = lambda pl: pl.run("global", -11.0, -1.0)
    else:
        A_w = 40
        S_w = rng.normal(0, 3, (A_w, A_w)).astype(np.float32)
        pw = []
        for L in lens:
            c = np.zeros((int(L), A_w), dtype=np.float32)
            for _ in range(4):
                c[np.arange(int(L)), rng.integers(0, A_w, int(L))] += rng.integers(1, 4, int(L))
            pw.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
        arena = nat.Arena(pw, S_w)
        run = lambda pl: pl.run("global", -11.0, -1.0)
    for wp in (False, True):
        plan = nat.Plan(arena, pairs, want_paths=wp)
        r = rate(plan, lambda: run(plan))
        print("%-5s paths=%d  %8.1f GCUPS  producer %d  %s" % (w, wp, r, plan.tile_producer(), plan.kernel_name()), flush=True)
        plan.close()
    arena.close()
    nat.set_match_mode(None)
    os.environ.pop("PRALINE_NO_REFTILE", None)
