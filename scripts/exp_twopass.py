"""Two-pass alignments with paths (flag-free forward fill + block recompute, dp_trace2.hip.h) against the single pass:
identical scores and paths on random batches (all modes, masks, float / one-hot / DNA), then the C3 and C2 rates."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix, nucleotide_matrix
from conftest import synth_profile
from bench import make_workload, synth_lengths, one_hot
nat.init(0)
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)

def run(arena, pairs, mode, rects, two):
    os.environ["PRALINE_TB_TWOPASS"] = "2" if two else "0"
    plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
    plan.run(mode, -11.0, -1.0)
    sc = plan.scores().copy()
    buf, off, rows = plan.paths_packed()
    name = plan.kernel_name()
    plan.close()
    return sc, buf, off, rows, name

n_cases = 0
t_end = time.time() + float(os.environ.get("SECONDS", "60"))
while time.time() < t_end:
    kind = rng.choice(["onehot", "profile", "dna"])
    N = int(rng.choice([2, 5, 17, 40, 90]))
    mu = int(rng.choice([3, 20, 40, 70, 130, 260, 520])) if kind != "dna" else int(rng.choice([50, 300, 900]))
    lens = np.maximum(1, rng.integers(max(1, mu // 2), mu * 3 // 2 + 1, N))
    if kind == "dna":
        S = nucleotide_matrix(); profs = [np.eye(15, dtype=np.float32)[rng.integers(0, 4, int(L))] for L in lens]
    elif kind == "onehot":
        S = blosum62_matrix(); profs = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
    else:
        S = blosum62_matrix(); profs = [synth_profile(rng, int(L))[0] for L in lens]
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < rng.choice([0.3, 1.0])]
    if len(pairs) == 0:
        continue
    mode = MODES[int(rng.integers(0, 5))]
    rects = None
    if mode == "local" and rng.random() < 0.5:
        rects = []
        for (i, j) in pairs:
            rr = []
            for _ in range(int(rng.integers(0, 4))):
                y0 = int(rng.integers(1, lens[i] + 1)); x0 = int(rng.integers(1, lens[j] + 1))
                rr.append((y0, min(int(lens[i]), y0 + int(rng.integers(0, 40))), x0, min(int(lens[j]), x0 + int(rng.integers(0, 40)))))
            rects.append(rr)
    arena = nat.Arena(profs, S)
    a = run(arena, pairs, mode, rects, False)
    b = run(arena, pairs, mode, rects, True)
    arena.close()
    assert "true>" in b[4].replace(" ", "") and a[4] != b[4], (a[4], b[4])
    if not np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)):
        k = int(np.flatnonzero(a[0] != b[0])[0])
        print("SCORE MISMATCH", kind, N, mu, mode, rects is not None, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], a[0][k], b[0][k]); sys.exit(1)
    for k in range(len(pairs)):
        pa = a[1][a[2][k]:a[2][k] + a[3][k]]; pb = b[1][b[2][k]:b[2][k] + b[3][k]]
        if not np.array_equal(pa, pb):
            print("PATH MISMATCH", kind, N, mu, mode, rects[k] if rects else None, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], len(pa), len(pb))
            d = min(len(pa), len(pb))
            bad = [q for q in range(1, d + 1) if not np.array_equal(pa[-q], pb[-q])]
            print(" first difference from the end at", bad[:1], pa[-(bad[0] if bad else 1)], pb[-(bad[0] if bad else 1)])
            sys.exit(1)
    n_cases += 1
print("two-pass == single pass on %d random batches" % n_cases)
os.environ.pop("PRALINE_TB_TWOPASS", None)
if os.environ.get("RATES", "1") == "1":
    S = blosum62_matrix()
    r3 = np.random.default_rng(3)
    l3 = synth_lengths(r3, 1024, 250)
    a3 = nat.Arena([one_hot(r3.integers(0, 20, int(L)), 27) for L in l3], S)
    i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
    p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
    c3 = int((l3[p3[:, 0]].astype(np.int64) * l3[p3[:, 1]]).sum())
    for two in ("0", "1"):
        os.environ["PRALINE_TB_TWOPASS"] = two
        for mode in ("global", "local"):
            pl = nat.Plan(a3, p3, want_paths=True)
            pl.run(mode, -11, -1); nat.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): pl.run(mode, -11, -1)
            nat.synchronize(); t1 = time.perf_counter()
            print("C3 twopass=%s %-8s %.2f ms  %.0f GCUPS  %s" % (two, mode, (t1 - t0) / 3 * 1e3, c3 * 3 / (t1 - t0) / 1e9, pl.kernel_name()), flush=True)
            pl.close()
    a3.close()
    w = make_workload("c2")
    ii, jj = np.triu_indices(256, k=1)
    pairs = np.stack([ii, jj], axis=1).astype(np.int32)
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    ar = nat.Arena(w["profs"], w["S"])
    for two in ("0", "2"):
        os.environ["PRALINE_TB_TWOPASS"] = two
        pl = nat.Plan(ar, pairs, want_paths=True)
        pl.run("global", -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): pl.run("global", -11, -1)
        nat.synchronize(); t1 = time.perf_counter()
        print("C2 float twopass=%s %.2f ms  %.0f GCUPS  %s" % (two, (t1 - t0) / 5 * 1e3, cells * 5 / (t1 - t0) / 1e9, pl.kernel_name()), flush=True)
        pl.close()
