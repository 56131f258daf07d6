"""Tasks cut between two pipeline workgroups (dp_types.h, PRALINE_PIPE_CUTS): kernel time and bitwise equality of the
scores against the whole-task schedule on C2-like batches."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def bits(a): return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
for N, mu in ((256, 400), (192, 400), (128, 400), (256, 250), (320, 300), (256, 700)):
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, N, mu)
    profs = [synth_profile(rng, int(L)) for L in lens]
    arena = nat.Arena(profs, S)
    pairs = np.stack(np.triu_indices(N, 1), axis=1).astype(np.int32)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    res = {}
    for cuts in ("0", "1"):
        os.environ["PRALINE_PIPE_CUTS"] = cuts
        plan = nat.Plan(arena, pairs)
        out = {}
        for mode in ("global", "local", "semiglobal_both"):
            ks = []
            for _ in range(12):
                plan.run(mode, -11.0, -1.0); nat.synchronize(); ks.append(plan.kernel_ms())
            out[mode] = (float(np.median(ks[2:])), plan.scores().copy())
        res[cuts] = (out, plan.cut_tasks, plan.kernel_name())
        plan.close()
    for mode in ("global", "local", "semiglobal_both"):
        a, b = res["0"][0][mode], res["1"][0][mode]
        same = np.array_equal(bits(a[1]), bits(b[1]))
        print("N=%d mu=%d %-16s whole %.3f ms (%.0f GCUPS)  cut %.3f ms (%.0f GCUPS, %d cuts)  equal=%s  %s" % (
            N, mu, mode, a[0], cells / a[0] / 1e6, b[0], cells / b[0] / 1e6, res["1"][1], same, res["1"][2]), flush=True)
    arena.close()
