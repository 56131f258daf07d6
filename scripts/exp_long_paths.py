"""Plans with paths whose packed traceback exceeds the scratch budget (long sequences): chain mode per chunk against task
mode (PRALINE_NO_CHAIN=1); scores and paths compared."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
def run(ar, pairs, mode, env):
    os.environ.pop("PRALINE_NO_CHAIN", None)
    os.environ.update(env)
    pl = nat.Plan(ar, pairs, want_paths=True)
    pl.run(mode, -11, -1); nat.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): pl.run(mode, -11, -1)
    nat.synchronize(); dt = (time.perf_counter() - t0) / 2
    sc = pl.scores().copy(); pk = pl.paths_packed(); kn = pl.kernel_name(); pl.close()
    return dt, sc, pk, kn
for N, mu, mode in ((256, 1000, "global"), (256, 1000, "local"), (128, 2500, "semiglobal_both"), (96, 5000, "global")):
    rng = np.random.default_rng(N + mu); lens = synth_lengths(rng, N, mu)
    profs = [synth_profile(rng, int(L)) for L in lens]
    ar = nat.Arena(profs, S)
    pairs = allpairs.enumerate_pairs(N)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    a = run(ar, pairs, mode, {}); b = run(ar, pairs, mode, {"PRALINE_NO_CHAIN": "1"})
    same = np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2][2], b[2][2])
    if same:
        for q in range(len(pairs)):
            if not np.array_equal(a[2][0][a[2][1][q]:a[2][1][q] + a[2][2][q]], b[2][0][b[2][1][q]:b[2][1][q] + b[2][2][q]]):
                same = False; break
    print("N=%d mu=%d %-16s pairs %6d | chain chunks %8.2f ms %5.0f GCUPS | task mode %8.2f ms %5.0f GCUPS | x%.2f | %s" % (
        N, mu, mode, len(pairs), a[0] * 1e3, cells / a[0] / 1e9, b[0] * 1e3, cells / b[0] / 1e9, b[0] / a[0], "scores and paths equal" if same else "DIFFER"), flush=True)
    ar.close()
