"""Per-rank kernel rate of the weak-scaling bench on ONE GPU: build the world-size-W batch, align one
rank's shard (column shards vs contiguous row-major slices)."""
import sys, os, math, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile, PAIRS_PER_GPU
nat.init(0)
S = blosum62_matrix()
for world in (1, 2, 4, 8):
    n = int(round(0.5 + math.sqrt(0.25 + 2.0 * world * PAIRS_PER_GPU)))
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, n, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    pairs = allpairs.enumerate_pairs(n)
    cells = lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]
    ar = nat.Arena(profs, S)
    b = allpairs.shard_bounds(cells, world)
    shards = allpairs.shard_columns(lens, pairs, world)
    for name, idxs in (("rows", [np.arange(b[r], b[r + 1]) for r in range(world)]), ("cols", shards)):
        rates = []
        for r in sorted(set([0, world // 2, world - 1])):
            pl = nat.Plan(ar, pairs[idxs[r]])
            for _ in range(2): pl.run("global", -11, -1)
            ms = []
            for _ in range(5):
                pl.run("global", -11, -1); ms.append(pl.kernel_ms())
            rates.append((r, pl.tasks, float(np.median(ms)), cells[idxs[r]].sum() / np.median(ms) / 1e6))
            pl.close()
        print("world=%d n=%d %s: " % (world, n, name) + "  ".join("rank %d: %d tasks %.2f ms %.0f GCUPS" % x for x in rates), flush=True)
    ar.close()
