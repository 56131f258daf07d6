"""Cycles per DP step of k_dp_split16: T identical tasks (32 pairs x L x L), T swept from one lone wave
to two waves per SIMD.  Separates single-wave latency from contention."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_profile
nat.init(0)
S = blosum62_matrix()
L = int(os.environ.get("L", "416"))
NSEQ = int(os.environ.get("NSEQ", "96"))
rng = np.random.default_rng(5)
if os.environ.get("ONEHOT", "0") == "1":
    profs = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, L)] for _ in range(NSEQ)]
else:
    profs = [synth_profile(rng, L) for _ in range(NSEQ)]
ar = nat.Arena(profs, S)
for T in [int(x) for x in os.environ.get("TS", "1,1024,2048,4096").split(",")]:
    pairs = []
    for t in range(T):
        two = 64 + t % 32
        base = (t // 32) % 2 * 32
        pairs += [(base + q, two) for q in range(32)]
    pairs = np.array(pairs, dtype=np.int32)
    pl = nat.Plan(ar, pairs)
    for _ in range(2): pl.run("global", -11, -1)
    ms = []
    for _ in range(5):
        pl.run("global", -11, -1); ms.append(pl.kernel_ms())
    ms = float(np.median(ms))
    steps = (L // 32) * (L + 17)
    waves_per_simd = max(1.0, T / 1024.0)
    print("T=%d kernel_ms=%.3f  us/step(per wave, serial)=%.3f  GCUPS=%.0f" % (
        T, ms, ms * 1e3 / (steps * np.ceil(T / 2048.0)), T * 32 * L * L / ms / 1e6), flush=True)
    pl.close()
ar.close()
