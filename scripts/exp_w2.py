"""Two-wave workgroups (k_dp_split16 W2): share threshold sweep on C2-like batches, and the large-batch
control (XCD group size) in the same process."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def run(N, env):
    for k in ("PRALINE_W_SNAKE", "PRALINE_W_SLOTS", "PRALINE_W2_FRAC", "PRALINE_NO_W2", "PRALINE_XCD_GROUP", "PRALINE_XCD_GROUP_W2"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, N, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S); pl = nat.Plan(ar, pairs)
    for _ in range(2): pl.run("global", -11, -1)
    ms = []
    for _ in range(7):
        pl.run("global", -11, -1); ms.append(pl.kernel_ms())
    print("N=%d %s kernel_ms=%.3f GCUPS=%.0f" % (N, env, np.median(ms), cells / np.median(ms) / 1e6), flush=True)
    pl.close(); ar.close()
for N in (32, 64, 128, 200, 256, 300, 340):
    run(N, {})
