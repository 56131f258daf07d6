"""Host-inclusive rate on C2: host profiles in -> arena (H2D + pre-multiply) -> plan (host scheduling + upload)
-> kernels -> scores back on the host."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix(); N = 256
rng = np.random.default_rng(2); lens = synth_lengths(rng, N, 400)
profs = [synth_profile(rng, int(L)) for L in lens]
pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
for rep in range(3):
    tc = time.perf_counter()
    cat = np.ascontiguousarray(np.concatenate(profs, axis=0), dtype=np.float32)
    print("np.concatenate of the profiles: %.2f ms (inside arena below)" % ((time.perf_counter() - tc) * 1e3))
    t0 = time.perf_counter()
    ar = nat.Arena(profs, S)
    t1 = time.perf_counter()
    pl = nat.Plan(ar, pairs)
    t2 = time.perf_counter()
    pl.run("global", -11, -1)
    sc = pl.scores()
    t3 = time.perf_counter()
    pl.close(); ar.close()
    print("rep %d: arena %.2f ms, plan %.2f ms, run+copy %.2f ms, total %.2f ms -> %.0f GCUPS host-inclusive" % (
        rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3, cells/(t3-t0)/1e9))
