"""The batched RawPairwiseAligner from HOST arrays (the operator path: BatchManager hands over numpy arrays): time of
native.RawBatch(requests) (staging + PCIe + layout), run and results for N requests of 400 x 400."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
nat.init(0)
rng = np.random.default_rng(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
reqs = []
for _ in range(N):
    m = (rng.standard_normal((400, 400)) * 3 - 0.5).astype(np.float32)
    g = [np.stack([-rng.uniform(5, 12, 400), -rng.uniform(0.5, 2, 400)], axis=1).astype(np.float32) for _ in range(2)]
    reqs.append((m, g[0], g[1], None))
for rep in range(3):
    t0 = time.perf_counter()
    rb = nat.RawBatch(reqs)
    t1 = time.perf_counter()
    rb.run("global")
    scores, paths = rb.results()
    t2 = time.perf_counter()
    rb.close()
    cells = N * 160000
    print("%d requests: create %.1f ms, run + scores + paths %.1f ms, total %.1f ms = %.1f GCUPS host to host (device %.3f ms)" % (
        N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, cells / (t2 - t0) / 1e9, rb.last_kernel_ms() if False else 0), flush=True)
