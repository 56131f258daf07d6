import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from praline_amd.matrices import nucleotide_matrix
from bench import synth_lengths
from oracle import oracle as orc
nat.init(0)
S = nucleotide_matrix()
N = int(os.environ.get("N", "96"))
rng = np.random.default_rng(5)
lens = synth_lengths(rng, N, 5000)
profs = []
for L in lens:
    p = np.zeros((L, 15), np.float32); p[np.arange(L), rng.integers(0, 4, L)] = 1; profs.append(p)
pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
t0 = time.perf_counter(); ar = nat.Arena(profs, S); t1 = time.perf_counter()
print("arena", ar.info(), "create %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); pl = nat.Plan(ar, pairs); t1 = time.perf_counter()
print("plan create %.1f ms, pairs %d, cells %.3g" % ((t1 - t0) * 1e3, len(pairs), cells))
for mode in ("global", "local", "semiglobal_both"):
    pl.run(mode, -11, -1); ms = pl.kernel_ms(); sc = pl.scores()
    print("%-16s kernel %.1f ms  %.0f GCUPS" % (mode, ms, cells / ms / 1e6))
    for k in (0, len(pairs) // 2, len(pairs) - 1):
        i, j = pairs[k]
        ref = orc.pairwise_score_fast(mode, profs[i], profs[j], S, -11.0, -1.0)
        assert sc[k] == np.float32(ref), (mode, i, j, sc[k], ref)
print("C5-like scores bit-identical to the oracle on the sampled pairs")
