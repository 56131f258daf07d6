"""One profile-profile alignment with paths (a merge step of the progressive MSA) repeated: device time per call;
run under rocprofv3 --kernel-trace --stats for the kernel split."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_profile
nat.init(0)
S = blosum62_matrix()
rng = np.random.default_rng(1)
for L in (350, 1000):
    profs = [synth_profile(rng, L), synth_profile(rng, L + 17)]
    ar = nat.Arena(profs, S)
    pl = nat.Plan(ar, np.array([(0, 1)], dtype=np.int32), want_paths=True)
    for mode in ("global", "semiglobal_both"):
        pl.run(mode, -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            pl.run(mode, -11, -1)
        nat.synchronize()
        print("L=%d %-16s %.3f ms per alignment (device pipeline, no copies)" % (L, mode, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
    pl.close(); ar.close()
