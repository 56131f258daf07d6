import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
rng = np.random.default_rng(4)
N = 4096
lens = synth_lengths(rng, N, 400)
pairs = allpairs.enumerate_pairs(N)
mine = pairs[allpairs.shard_columns(lens, pairs, 8)[3]]
if os.environ.get("ONEHOT") == "1":   # plain sequences: integer scoring, the match-score lookup kernel
    from bench import one_hot
    profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
else:
    profs = [synth_profile(rng, int(L)) for L in lens]
ar = nat.Arena(profs, blosum62_matrix())
cells = int((lens[mine[:, 0]].astype(np.int64) * lens[mine[:, 1]]).sum())
for G in os.environ.get("GS", "0,16,64,256,1024,4096").split(","):
    os.environ["PRALINE_XCD_GROUP"] = G
    pl = nat.Plan(ar, mine)
    pl.run("global", -11, -1)
    ms = []
    for _ in range(3):
        pl.run("global", -11, -1); ms.append(pl.kernel_ms())
    print("G=%s tasks=%d %.1f ms %.0f GCUPS %s" % (G, pl.tasks, np.median(ms), cells / np.median(ms) / 1e6, pl.kernel_name()), flush=True)
    pl.close()
