import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix(); N = int(os.environ.get("N", "256"))
rng = np.random.default_rng(2); lens = synth_lengths(rng, N, 400); profs = [synth_profile(rng, int(L)) for L in lens]
pairs = np.array([(i, j) for i in range(N) for j in range(i + 1, N)], dtype=np.int32)
cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
ar = nat.Arena(profs, S); pl = nat.Plan(ar, pairs)
for _ in range(2): pl.run("global", -11, -1)
ms = []
for _ in range(5):
    pl.run("global", -11, -1); ms.append(pl.kernel_ms())
print("EXP=%%s N=%%d kernel_ms=%%.3f GCUPS=%%.0f" %% (os.environ.get("PRALINE_EXP", "0"), N, np.median(ms), cells / np.median(ms) / 1e6))
''' % ROOT
names = {0: "full", 1: "no B reload", 2: "no boundary ld/st", 3: "no B, no bnd", 4: "no MFMA", 8: "no DP", 7: "no mem, no MFMA", 11: "no mem, no DP", 12: "no MFMA no DP"}
for N in (256, 512):
    for e in [int(x) for x in os.environ.get("EXPS", "0,1,2,3,4,8,7,11,12").split(",")]:
        env = dict(os.environ, PRALINE_EXP=str(e), N=str(N))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print((out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], "   <-", names.get(e, ""), flush=True)
