"""Scores-only and with-paths rates over batch shapes (all pairs of N sequences of ~mu residues), float profiles and
plain sequences: looks for cliffs between the kernels' regimes."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile, one_hot
nat.init(0)
S = blosum62_matrix()
def rate(ar, pairs, cells, mode, paths):
    pl = nat.Plan(ar, pairs, want_paths=paths)
    pl.run(mode, -11, -1); nat.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps): pl.run(mode, -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / reps
    kn = pl.kernel_name(); pl.close()
    return "%8.2f ms %5.0f [%s]" % (dt * 1e3, cells / dt / 1e9, kn[:30])
shapes = [(4096, 30), (2048, 60), (1024, 120), (512, 250), (256, 500), (128, 1000), (64, 2500), (32, 5000), (16, 10000), (8, 20000), (2, 30000)]
for N, mu in shapes:
    rng = np.random.default_rng(N); lens = synth_lengths(rng, N, mu)
    pairs = allpairs.enumerate_pairs(N)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    for kind in ("float", "onehot"):
        if kind == "float": profs = [synth_profile(rng, int(L)) for L in lens]
        else: profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
        ar = nat.Arena(profs, S)
        print("N=%5d mu=%6d %-6s pairs %8d | scores %s | paths %s | local paths %s" % (N, mu, kind, len(pairs), rate(ar, pairs, cells, "global", False),
              rate(ar, pairs, cells, "global", True), rate(ar, pairs, cells, "local", True)), flush=True)
        ar.close()
