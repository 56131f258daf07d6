"""A/B of two library builds on the same box: PRALINE_LIB=<path> selects the build; prints C2 and a C4-rank-share
kernel rate for float profiles and one-hot (global, local)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def rate(ar, pairs, lens, mode, reps=5):
    pl = nat.Plan(ar, pairs); pl.run(mode, -11, -1)
    ms = []
    for _ in range(reps):
        pl.run(mode, -11, -1); ms.append(pl.kernel_ms())
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum()); pl.close()
    return cells / float(np.median(ms)) / 1e6
out = []
for name, N, seed, shard in (("C2", 256, 2, None), ("C4/8", 4096, 4, 3)):
    rng = np.random.default_rng(seed); lens = synth_lengths(rng, N, 400)
    pairs = allpairs.enumerate_pairs(N)
    if shard is not None: pairs = pairs[allpairs.shard_columns(lens, pairs, 8)[shard]]
    for kind in ("profile", "onehot"):
        profs = [synth_profile(rng, int(L)) if kind == "profile" else np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
        ar = nat.Arena(profs, S)
        for mode in ("global", "local"):
            out.append("%s %s %s %.0f" % (name, kind, mode, rate(ar, pairs, lens, mode)))
        ar.close()
print(os.environ.get("PRALINE_LIB", "default"), " | ".join(out), flush=True)
