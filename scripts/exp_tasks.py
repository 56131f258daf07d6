import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
for N in (128, 180, 230, 240, 256, 300, 362, 512):
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, N, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S); pl = nat.Plan(ar, pairs)
    ntasks = sum((j + 31) // 32 for j in range(N))
    for _ in range(2): pl.run("global", -11, -1)
    ms = []
    for _ in range(5):
        pl.run("global", -11, -1); ms.append(pl.kernel_ms())
    print("N=%d tasks~%d cells=%.3g kernel_ms=%.3f GCUPS=%.0f" % (N, ntasks, cells, np.median(ms), cells / np.median(ms) / 1e6))
    pl.close(); ar.close()
