"""k_dp_pk16_tb (two pairs per lane in packed int16, integer scoring) against k_dp_quad_tb (PRALINE_TB_PK16=0): identical scores
and paths on random one-hot batches - five modes, zero rectangles, integer / half-integer / odd gap scores (the latter fall
back to the strip kernels) -, then the rates on C2 one-hot and a C3 slice."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import make_workload, one_hot, synth_lengths
nat.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]

def run(arena, pairs, mode, quad, rects, gaps):
    os.environ["PRALINE_TB_PK16"] = "1" if quad else "0"
    plan = nat.Plan(arena, pairs, want_paths=True, rects=rects)
    plan.run(mode, *gaps)
    sc = plan.scores().copy()
    buf, off, rows = plan.paths_packed()
    name = plan.kernel_name()
    plan.close()
    return sc, buf, off, rows, name

n_cases = 0
t_end = time.time() + float(os.environ.get("SECONDS", "60"))
t_print = time.time()
S = blosum62_matrix()
while time.time() < t_end:
    N = int(rng.choice([2, 5, 17, 40, 70]))
    mu = int(rng.choice([1, 3, 9, 20, 40, 70, 130, 260]))
    lens = np.maximum(1, rng.integers(max(1, mu // 2), mu * 3 // 2 + 1, N))
    profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    allp = np.array([(i, j) for i in range(N) for j in range(N)], dtype=np.int32)
    pairs = allp[rng.random(len(allp)) < rng.choice([0.3, 1.0])]
    if len(pairs) == 0:
        continue
    mode = MODES[int(rng.integers(0, len(MODES)))]
    gaps = [(-11.0, -1.0), (-7.5, -0.5), (-10.3, -1.7), (-4.0, -4.0)][int(rng.integers(0, 4))]
    rects = None
    rk = rng.random()
    if rk < 0.5:
        nmax = 3 if rk < 0.35 else 7
        rects = []
        for (i, j) in pairs:
            rl = []
            for _ in range(int(rng.integers(0, nmax + 1))):
                y0 = int(rng.integers(1, lens[i] + 1)); x0 = int(rng.integers(1, lens[j] + 1))
                rl.append((y0, min(int(lens[i]), y0 + int(rng.integers(0, 20))), x0, min(int(lens[j]), x0 + int(rng.integers(0, 40)))))
            rects.append(rl)
    rects_many = rects is not None and max(len(r) for r in rects) > 4
    arena = nat.Arena(profs, S)
    a = run(arena, pairs, mode, False, rects, gaps)
    b = run(arena, pairs, mode, True, rects, gaps)
    arena.close()
    assert b[4].startswith("k_dp_pk16_tb") or gaps[0] != int(gaps[0]) or gaps == (-10.3, -1.7) or rects_many, (b[4], gaps)
    assert not a[4].startswith("k_dp_pk16_tb"), a[4]
    if not np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)):
        k = int(np.flatnonzero(a[0].view(np.uint32) != b[0].view(np.uint32))[0])
        print("SCORE MISMATCH", N, mu, mode, gaps, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], a[0][k], b[0][k], "rects" if rects else "", a[4]); sys.exit(1)
    for k in range(len(pairs)):
        pa = a[1][a[2][k]:a[2][k] + a[3][k]]; pb = b[1][b[2][k]:b[2][k] + b[3][k]]
        if not np.array_equal(pa, pb):
            print("PATH MISMATCH", N, mu, mode, gaps, pairs[k], lens[pairs[k][0]], lens[pairs[k][1]], len(pa), len(pb), "rects" if rects else "", a[4])
            d = min(len(pa), len(pb))
            bad = [q for q in range(1, d + 1) if not np.array_equal(pa[-q], pb[-q])]
            print(" first difference from the end at", bad[:1], pa[-(bad[0] if bad else 1)], pb[-(bad[0] if bad else 1)])
            sys.exit(1)
    n_cases += 1
    if time.time() - t_print > 30:
        print("  ...", n_cases, "batches", flush=True); t_print = time.time()
print("packed int16 == float kernels on %d random batches" % n_cases, flush=True)
if os.environ.get("RATES", "1") == "1":
    def rate(tag, arena, pairs, cells, modes):
        for mode in modes:
            for quad in ("0", "1"):
                os.environ["PRALINE_TB_PK16"] = quad
                plan = nat.Plan(arena, pairs, want_paths=True)
                for _ in range(2):
                    plan.run(mode, -11.0, -1.0)
                nat.synchronize()
                t = time.perf_counter()
                for _ in range(3):
                    plan.run(mode, -11.0, -1.0)
                nat.synchronize()
                dt = (time.perf_counter() - t) / 3
                print("%s %-16s PK16=%s  %-28s %.3f ms  %.0f GCUPS" % (tag, mode, quad, plan.kernel_name(), dt * 1e3, cells / dt / 1e9), flush=True)
                plan.close()
    w = make_workload("c2")
    rng2 = np.random.default_rng(2)
    oh = [one_hot(rng2.integers(0, 20, int(L)), 27) for L in w["lens"]]
    arena = nat.Arena(oh, w["S"])
    n = len(w["lens"])
    pairs = np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
    cells = int((w["lens"][pairs[:, 0]].astype(np.int64) * w["lens"][pairs[:, 1]]).sum())
    rate("C2 one-hot", arena, pairs, cells, ["global", "local"])
    arena.close()
    rng3 = np.random.default_rng(3)
    l3 = synth_lengths(rng3, 1024, 250)
    a3 = nat.Arena([one_hot(rng3.integers(0, 20, int(L)), 27) for L in l3], w["S"])
    i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
    p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
    if os.environ.get("C3_ALL") != "1":
        p3 = p3[p3[:, 1] % 8 == 3]
    c3 = int((l3[p3[:, 0]].astype(np.int64) * l3[p3[:, 1]]).sum())
    rate("C3 %d pairs" % len(p3), a3, p3, c3, ["global", "local", "semiglobal_both"])
    a3.close()
