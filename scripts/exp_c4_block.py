"""One rank's share of C4 (column shard 3 of 8, float profiles): k_dp_pipe kernel time against the scheduler's block size
(PRALINE_PIPE_BLOCK: 8 3567, 16 3526, 32 3445, 64 3258 GCUPS) and the workgroup slots it assumes (PRALINE_PIPE_SLOTS: flat)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
rng = np.random.default_rng(4); lens = synth_lengths(rng, 4096, 400)
pairs = allpairs.enumerate_pairs(4096)
pairs = pairs[allpairs.shard_columns(lens, pairs, 8)[3]]
profs = [synth_profile(rng, int(L)) for L in lens]
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
ar = nat.Arena(profs, S)
for block, slots in ((16, 512), (16, 256), (16, 1024), (16, 2048), (16, 128), (16, 512)):
    os.environ["PRALINE_PIPE_BLOCK"] = str(block); os.environ["PRALINE_PIPE_SLOTS"] = str(slots)
    pl = nat.Plan(ar, pairs); pl.run("global", -11, -1)
    ms = []
    for _ in range(4):
        pl.run("global", -11, -1); ms.append(pl.kernel_ms())
    print("block %2d slots %4d: tasks %6d steps %10d kernel %.2f ms %.0f GCUPS" % (block, slots, pl.tasks, pl.steps, float(np.median(ms)), cells / float(np.median(ms)) / 1e6), flush=True)
    pl.close()
