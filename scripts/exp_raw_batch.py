"""Batched RawPairwiseAligner (praline_raw_batch_*, k_rawb_fill): random requests against the oracle, then throughput on
N requests of 400 x 400 resident in HBM."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from oracle import oracle as orc
nat.init(0)
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]

def rand_requests(rng, n, lo, hi, zero_share=0.3, integer=False):
    reqs = []
    for _ in range(n):
        L1, L2 = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
        if integer:
            m = rng.integers(-4, 12, (L1, L2)).astype(np.float32)
        else:
            m = (rng.standard_normal((L1, L2)) * 3 - 0.5).astype(np.float32)
        g1 = np.stack([-rng.uniform(5, 12, L1), -rng.uniform(0.5, 2, L1)], axis=1).astype(np.float32)
        g2 = np.stack([-rng.uniform(5, 12, L2), -rng.uniform(0.5, 2, L2)], axis=1).astype(np.float32)
        z = None
        if rng.random() < zero_share:
            k = int(rng.integers(1, 40))
            z = [(int(rng.integers(0, L1 + 1)), int(rng.integers(0, L2 + 1))) for _ in range(k)]
        reqs.append((m, g1, g2, z))
    return reqs

def check(reqs, modes, tag):
    rb = nat.RawBatch(reqs)
    rb.run(modes)
    scores, paths = rb.results()
    bad = 0
    for r, (m, g1, g2, z) in enumerate(reqs):
        mo = modes if isinstance(modes, str) else modes[r]
        s, p = orc.raw_pairwise_align(mo, m, g1, g2, z)
        ok = np.float32(s) == scores[r] and np.array_equal(np.asarray(p, dtype=np.int64), paths[r].astype(np.int64))
        if not ok:
            bad += 1
            if bad <= 5:
                print("  MISMATCH", tag, r, mo, m.shape, "score", s, scores[r], "rows", len(p), len(paths[r]), flush=True)
    print("%s: %d requests, %d mismatches, kernel %.3f ms" % (tag, len(reqs), bad, rb.last_kernel_ms()), flush=True)
    rb.close()
    return bad

if __name__ == "__main__":
    rng = np.random.default_rng(11)
    total_bad = 0
    if "--bench-only" not in sys.argv:
        for mo in MODES:
            total_bad += check(rand_requests(rng, 60, 1, 90), mo, "small " + mo)
        total_bad += check(rand_requests(rng, 100, 50, 300, integer=True), [MODES[i % 5] for i in range(100)], "mixed integer")
        total_bad += check(rand_requests(rng, 40, 100, 700), [MODES[i % 5] for i in range(40)], "mixed float")
        total_bad += check([rand_requests(rng, 1, 1300, 1400)[0], rand_requests(rng, 1, 600, 3000)[0]], ["local", "global"], "long")
        print("mismatches:", total_bad, flush=True)
    if total_bad == 0 and "--no-bench" not in sys.argv:
        sizes = ((500, 400), (2048, 400), (256, 1000))
        if "--sizes" in sys.argv:
            sizes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[sys.argv.index("--sizes") + 1].split(",")]
        for n, L in sizes:
            reqs = rand_requests(rng, 8, L, L, zero_share=0.0)
            reqs = [reqs[i % 8] for i in range(n)]
            rb = nat.RawBatch(reqs)
            for mo in ("global", "local"):
                rb.run(mo); rb.results(paths=False)
                t0 = time.perf_counter()
                for _ in range(3): rb.run(mo)
                rb.results(paths=False)
                dt = (time.perf_counter() - t0) / 3
                print("%d x %dx%d %s: %.3f ms host-timed, %.3f ms device, %.0f GCUPS" % (n, L, L, mo, dt * 1e3, rb.last_kernel_ms(), rb.cells / dt / 1e9), flush=True)
            rb.close()
