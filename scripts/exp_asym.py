"""Asymmetric batches: one long sequence against many short ones, in both roles; scores-only and with paths."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_profile, one_hot
nat.init(0)
S = blosum62_matrix()
def rate(ar, pairs, cells, mode, paths):
    pl = nat.Plan(ar, pairs, want_paths=paths)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): pl.run(mode, -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / 3
    kn = pl.kernel_name(); t = pl.tasks; pl.close()
    return "%8.2f ms %5.0f GCUPS tasks %5d [%s]" % (dt * 1e3, cells / dt / 1e9, t, kn[:34])
rng = np.random.default_rng(1)
for n_short, l_short, l_long in ((2000, 100, 5000), (500, 300, 20000), (5000, 50, 1000)):
    lens = np.array([l_long] + [l_short] * n_short)
    for kind in ("float", "onehot"):
        profs = [synth_profile(rng, int(L)) for L in lens] if kind == "float" else [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
        ar = nat.Arena(profs, S)
        for role, pairs in (("long as two", np.array([(k, 0) for k in range(1, n_short + 1)], dtype=np.int32)),
                            ("long as one", np.array([(0, k) for k in range(1, n_short + 1)], dtype=np.int32))):
            cells = int(n_short) * l_short * l_long
            print("%d x %d vs %d %-6s %-11s | scores %s | paths %s" % (n_short, l_short, l_long, kind, role, rate(ar, pairs, cells, "global", False), rate(ar, pairs, cells, "global", True)), flush=True)
        ar.close()
