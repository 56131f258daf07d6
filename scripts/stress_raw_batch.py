"""Random batches of RawPairwiseAligner requests against the oracle for a given time: shapes from 1 x 1 to 700 x 900 (now and
then a request of more than 512 rows or 2 600 columns), float and integer scores (ties), per-position gap scores, zero cells,
modes mixed within a batch.  usage: scripts/stress_raw_batch.py [seconds] [seed]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from oracle import oracle as orc
nat.init(0)
MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
batches = alignments = cells = 0
while time.time() < t_end:
    n = int(rng.integers(1, 80))
    hi = int(rng.choice([8, 40, 130, 300, 700]))
    reqs, modes = [], []
    for _ in range(n):
        L1, L2 = int(rng.integers(1, hi + 1)), int(rng.integers(1, hi + 200 if hi == 700 else hi + 1))
        if rng.random() < 0.02: L1 = int(rng.integers(513, 1500))
        if rng.random() < 0.01: L2 = int(rng.integers(2600, 3400)); L1 = min(L1, 200)
        kind = rng.integers(0, 3)
        if kind == 0: m = rng.integers(-4, 12, (L1, L2)).astype(np.float32)
        elif kind == 1: m = (rng.standard_normal((L1, L2)) * 3 - 0.5).astype(np.float32)
        else: m = np.where(rng.random((L1, L2)) < 0.1, 5.0, -1.0).astype(np.float32)
        if rng.random() < 0.5:
            g1 = np.stack([-rng.uniform(5, 12, L1), -rng.uniform(0.5, 2, L1)], axis=1).astype(np.float32)
            g2 = np.stack([-rng.uniform(5, 12, L2), -rng.uniform(0.5, 2, L2)], axis=1).astype(np.float32)
        else:
            g1 = np.tile(np.array([[-11.0, -1.0]], np.float32), (L1, 1)); g2 = np.tile(np.array([[-11.0, -1.0]], np.float32), (L2, 1))
        if kind == 0: g1, g2 = np.round(g1), np.round(g2)
        z = None
        r = rng.random()
        if r < 0.25:
            z = [(int(rng.integers(0, L1 + 1)), int(rng.integers(0, L2 + 1))) for _ in range(int(rng.integers(1, 50)))]
        elif r < 0.35:     # a rectangle, as Waterman-Eggert masks are
            y0, x0 = int(rng.integers(1, L1 + 1)), int(rng.integers(1, L2 + 1))
            z = [(y, x) for y in range(y0, min(L1, y0 + 12) + 1) for x in range(x0, min(L2, x0 + 12) + 1)]
        reqs.append((m, g1, g2, z)); modes.append(MODES[int(rng.integers(0, 5))])
    rb = nat.RawBatch(reqs)
    scores, paths = rb.run(modes).results()
    rb.close()
    for r, (m, g1, g2, z) in enumerate(reqs):
        s, p = orc.raw_pairwise_align(modes[r], m, g1, g2, z)
        if not (np.float32(s) == scores[r] and np.array_equal(np.asarray(p), paths[r])):
            print("MISMATCH batch %d request %d mode %s shape %s seed %d" % (batches, r, modes[r], m.shape, seed), flush=True)
            np.savez("gpurun_out/stress_raw_fail.npz", m=m, g1=g1, g2=g2, z=np.array(z if z else []), mode=modes[r])
            sys.exit(1)
        alignments += 1; cells += m.size
    batches += 1
    if batches % 50 == 0: print("%d batches, %d alignments, %.3g cells" % (batches, alignments, cells), flush=True)
print("stress ok: %d batches, %d alignments, %.3g cells, seed %d" % (batches, alignments, cells, seed), flush=True)
