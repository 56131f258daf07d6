"""Kernel rate of k_dp_pipe against k_dp_split16 (PRALINE_NO_PIPE=1) over batch sizes: all pairs of N float-profile
sequences of ~400 aa (global), plus one rank's share of C4."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def rate(ar, pairs, lens, mode, pipe, reps=5):
    if pipe: os.environ.pop("PRALINE_NO_PIPE", None)
    else: os.environ["PRALINE_NO_PIPE"] = "1"
    pl = nat.Plan(ar, pairs); pl.run(mode, -11, -1)
    ms = []
    for _ in range(reps):
        pl.run(mode, -11, -1); ms.append(pl.kernel_ms())
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum()); kn = pl.kernel_name(); pl.close()
    return cells / float(np.median(ms)) / 1e6, float(np.median(ms)), kn
for N in [int(x) for x in os.environ.get("NS", "32,64,128,192,256,300,362,512,724,1024").split(",")]:
    rng = np.random.default_rng(N); lens = synth_lengths(rng, N, int(os.environ.get("MU", "400")))
    profs = [synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    pairs = allpairs.enumerate_pairs(N)
    for mode in os.environ.get("MODES", "global").split(","):
        a = rate(ar, pairs, lens, mode, True); b = rate(ar, pairs, lens, mode, False)
        print("N=%5d %-8s pairs %8d  pipe %8.3f ms %6.0f GCUPS | tasks %8.3f ms %6.0f GCUPS  (x%.2f)  [%s | %s]" % (N, mode, len(pairs), a[1], a[0], b[1], b[0], a[0] / b[0], a[2], b[2]), flush=True)
    ar.close()
