import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def run(N, kind, shard=None):
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, N, 400)
    profs = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] if kind == "onehot" else synth_profile(rng, int(L)) for L in lens]
    pairs = allpairs.enumerate_pairs(N)
    if shard: pairs = pairs[allpairs.shard_columns(lens, pairs, shard)[3]]
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S)
    out = []
    for env in ("0", "1"):
        os.environ["PRALINE_ALL_SINGLES"] = env
        pl = nat.Plan(ar, pairs)
        pl.run("global", -11, -1)
        ms = []
        for _ in range(3):
            pl.run("global", -11, -1); ms.append(pl.kernel_ms())
        out.append("singles=%s %.1f ms %.0f GCUPS" % (env, np.median(ms), cells / np.median(ms) / 1e6))
        t = pl.tasks
        pl.close()
    print("N=%d %s tasks=%d: %s" % (N, kind, t, " | ".join(out)), flush=True)
    ar.close()
for N in (362, 512, 1024):
    run(N, "profile"); run(N, "onehot")
run(4096, "profile", 8); run(4096, "onehot", 8)
