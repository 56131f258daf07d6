import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
for N in (362, 512, 724, 1024, 1448):
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, N, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    pairs = allpairs.enumerate_pairs(N)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S)
    out = []
    for G in ("16", ""):
        if G: os.environ["PRALINE_XCD_GROUP"] = G
        else: os.environ.pop("PRALINE_XCD_GROUP", None)
        pl = nat.Plan(ar, pairs)
        pl.run("global", -11, -1)
        ms = []
        for _ in range(3):
            pl.run("global", -11, -1); ms.append(pl.kernel_ms())
        out.append("G=%s %.1f ms %.0f GCUPS" % (G or "auto", np.median(ms), cells / np.median(ms) / 1e6))
        t = pl.tasks
        pl.close()
    print("N=%d tasks=%d: %s" % (N, t, " | ".join(out)), flush=True)
    ar.close()
