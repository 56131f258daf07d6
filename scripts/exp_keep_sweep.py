"""Float profiles, global mode, with paths: the default path plans (chain / task mode) against the kept-state forward
fill on the scores kernel + block recompute (PRALINE_TB_KEEP=1), over batch sizes."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def run(ar, pairs, env):
    for k in ("PRALINE_TB_KEEP",): os.environ.pop(k, None)
    os.environ.update(env)
    pl = nat.Plan(ar, pairs, want_paths=True)
    pl.run("global", -11, -1); nat.synchronize()
    reps = 5 if len(pairs) < 100000 else 2
    t0 = time.perf_counter()
    for _ in range(reps): pl.run("global", -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / reps
    kn = pl.kernel_name(); sc = pl.scores().copy(); pl.close()
    return dt, kn, sc
for N, mu in ((2, 400), (8, 400), (24, 400), (64, 400), (128, 400), (192, 400), (256, 400), (384, 400), (512, 400), (724, 400), (256, 150), (256, 1000)):
    rng = np.random.default_rng(N + mu); lens = synth_lengths(rng, N, mu)
    profs = [synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    pairs = allpairs.enumerate_pairs(N)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    a = run(ar, pairs, {"PRALINE_TB_KEEP": "0"}); b = run(ar, pairs, {"PRALINE_TB_KEEP": "1"})
    print("N=%4d mu=%4d pairs %7d | default %8.3f ms %5.0f GCUPS [%s] | keep %8.3f ms %5.0f GCUPS [%s] x%.2f %s" % (
        N, mu, len(pairs), a[0] * 1e3, cells / a[0] / 1e9, a[1][:28], b[0] * 1e3, cells / b[0] / 1e9, b[1][:36], a[0] / b[0],
        "" if np.array_equal(a[2], b[2]) else "SCORES DIFFER"), flush=True)
    ar.close()
