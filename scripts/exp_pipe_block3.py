"""Scheduler block size (PRALINE_PIPE_BLOCK: sequences two per block) for plans larger than the resident workgroup
slots: one rank's share of C4, all of C4, and all pairs of 512 / 1024 sequences ~400 aa - kernel time and plan creation."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
import bench
nat.init(0)
w = bench.make_workload("c4")
lens = np.asarray(w["lens"])
arena = nat.Arena(w["profs"], w["S"])
iu = allpairs.enumerate_pairs(4096)
share = iu[allpairs.shard_columns(lens, iu, 8)[3]]
sub = lambda n: np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32)
for tag, pairs in (("C4 share 1/8", share), ("512 seqs all pairs", sub(512)), ("1024 seqs all pairs", sub(1024)), ("C4 all pairs", iu)):
    cells = float((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ref = None
    for block in ("16", "8", "12", "16", "8"):
        os.environ["PRALINE_PIPE_BLOCK"] = block
        t0 = time.perf_counter(); pl = nat.Plan(arena, pairs); t1 = time.perf_counter()
        pl.run("global", -11.0, -1.0)
        ms = []
        for _ in range(3):
            pl.run("global", -11.0, -1.0); nat.synchronize(); ms.append(pl.kernel_ms())
        sc = pl.scores()
        if ref is None: ref = sc
        same = bool(np.array_equal(ref.view(np.uint32), sc.view(np.uint32)))
        print("%-20s block %2s: plan %.1f ms, kernel %.2f ms %.0f GCUPS, same scores %s" % (tag, block, (t1 - t0) * 1e3, float(np.median(ms)), cells / float(np.median(ms)) / 1e6, same), flush=True)
        pl.close()
