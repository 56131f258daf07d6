"""Kernel rates on the BASELINE configurations' shapes (one GPU): C4 = one rank's column shard of
4096 seqs ~400 aa (1/8 of 8.4 M pairs), C5 = a 1/14 column shard of 512 DNA seqs ~5 kb."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix, nucleotide_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
def rate(ar, pairs, lens, mode="global", reps=3):
    pl = nat.Plan(ar, pairs)
    pl.run(mode, -11, -1)
    ms = []
    for _ in range(reps):
        pl.run(mode, -11, -1); ms.append(pl.kernel_ms())
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    t = pl.tasks
    pl.close()
    return cells, float(np.median(ms)), t
rng = np.random.default_rng(4)
N = 4096
lens = synth_lengths(rng, N, 400)
pairs = allpairs.enumerate_pairs(N)
mine = pairs[allpairs.shard_columns(lens, pairs, 8)[3]]
S = blosum62_matrix()
for kind in ("onehot", "profile"):
    profs = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] if kind == "onehot" else synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    for mode in ("global", "local", "semiglobal_both"):
        cells, ms, t = rate(ar, mine, lens, mode)
        print("C4 rank share %-8s %-16s pairs=%d tasks=%d cells=%.3g  %.1f ms  %.0f GCUPS" % (kind, mode, len(mine), t, cells, ms, cells / ms / 1e6), flush=True)
    ar.close(); del profs
rng = np.random.default_rng(5)
N = 512
lens = synth_lengths(rng, N, 5000)
profs = [np.eye(15, dtype=np.float32)[rng.integers(0, 4, int(L))] for L in lens]
pairs = allpairs.enumerate_pairs(N)
ar = nat.Arena(profs, nucleotide_matrix())
for frac in (14, 4):
    mine = pairs[allpairs.shard_columns(lens, pairs, frac)[1]]
    cells, ms, t = rate(ar, mine, lens, "global", reps=2)
    print("C5 1/%d shard onehot global pairs=%d tasks=%d cells=%.3g  %.1f ms  %.0f GCUPS" % (frac, len(mine), t, cells, ms, cells / ms / 1e6), flush=True)
ar.close()
