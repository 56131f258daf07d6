"""Score plans of a few long sequences: chain mode without flags (PRALINE_SCORES_CHAIN=1) against the shared-wave score
kernels (=0), scores compared bitwise in all five modes, and what the built-in estimate picks."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile, one_hot
nat.init(0)
S = blosum62_matrix()
MODES = ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two")
def run(ar, pairs, mode, env):
    os.environ.pop("PRALINE_SCORES_CHAIN", None)
    if env is not None: os.environ["PRALINE_SCORES_CHAIN"] = env
    pl = nat.Plan(ar, pairs)
    pl.run(mode, -11, -1); nat.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps): pl.run(mode, -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / reps
    sc = pl.scores().copy(); kn = pl.kernel_name(); pl.close()
    return dt, sc, kn
shapes = [(2, 400), (2, 1000), (2, 3000), (2, 10000), (4, 700), (8, 400), (8, 2000), (16, 300), (16, 1000), (24, 400), (32, 1500), (48, 600), (64, 400), (64, 2500), (100, 400)]
for N, mu in shapes:
    for kind in ("float", "onehot"):
        rng = np.random.default_rng(N * 7 + mu); lens = synth_lengths(rng, N, mu)
        profs = [synth_profile(rng, int(L)) for L in lens] if kind == "float" else [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
        ar = nat.Arena(profs, S)
        pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32) if N <= 4 else allpairs.enumerate_pairs(N)
        cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
        same = True
        for mode in MODES:
            a = run(ar, pairs, mode, "1"); b = run(ar, pairs, mode, "0")
            same = same and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
            if mode == "global": ta, tb, ka, kb = a[0], b[0], a[2], b[2]
        auto = run(ar, pairs, "global", None)
        pick = "chain" if "true, true" in auto[2] else "shared"
        print("N=%4d mu=%6d %-6s pairs %6d | chain %8.3f ms %5.0f GCUPS | shared %8.3f ms %5.0f GCUPS | x%5.2f | auto picks %-6s %s | %s" % (
            N, mu, kind, len(pairs), ta * 1e3, cells / ta / 1e9, tb * 1e3, cells / tb / 1e9, tb / ta, pick,
            "ok" if (pick == "chain") == (ta < tb) else "WRONG PICK", "all modes bitwise equal" if same else "SCORES DIFFER"), flush=True)
        ar.close()
