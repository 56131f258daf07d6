"""k_dp_pipe kernel time over batch sizes for several scheduler block sizes (PRALINE_PIPE_BLOCK), XCD placement off."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
os.environ["PRALINE_PIPE_XCD"] = "0"
for N in [int(x) for x in os.environ.get("NS", "128,160,192,224,256,288,320,384,512,724,1024").split(",")]:
    rng = np.random.default_rng(N); lens = synth_lengths(rng, N, int(os.environ.get("MU", "400")))
    profs = [synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    pairs = allpairs.enumerate_pairs(N)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    out = []
    for block in (8, 16, 32, 64):
        os.environ["PRALINE_PIPE_BLOCK"] = str(block)
        pl = nat.Plan(ar, pairs); pl.run("global", -11, -1)
        ms = []
        for _ in range(7):
            pl.run("global", -11, -1); ms.append(pl.kernel_ms())
        out.append("b%-2d %7.3f ms %5.0f%s" % (block, float(np.median(ms)), cells / float(np.median(ms)) / 1e6, "" if "pipe" in pl.kernel_name() else "*"))
        pl.close()
    print("N=%5d pairs %7d | %s" % (N, len(pairs), " | ".join(out)), flush=True)
    ar.close()
