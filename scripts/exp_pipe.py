"""k_dp_pipe (pipeline workgroups) against k_dp_split16 (PRALINE_NO_PIPE=1): scores compared bitwise, kernel times.
  python scripts/exp_pipe.py [quick]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
MODES = ("global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two")

def run(ar, pairs, mode, pipe, reps):
    if pipe: os.environ.pop("PRALINE_NO_PIPE", None)
    else: os.environ["PRALINE_NO_PIPE"] = "1"
    pl = nat.Plan(ar, pairs)
    pl.run(mode, -11, -1)
    sc = pl.scores().copy()
    ms = []
    for _ in range(reps):
        pl.run(mode, -11, -1); ms.append(pl.kernel_ms())
    kn = pl.kernel_name(); pl.close()
    return sc, (float(np.median(ms)) if ms else 0.0), kn

def check(name, lens, pairs, profs, modes=MODES, reps=0):
    ar = nat.Arena(profs, S)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    for mode in modes:
        a, ta, ka = run(ar, pairs, mode, True, reps)
        b, tb, kb = run(ar, pairs, mode, False, reps)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        msg = "%-22s %-16s pairs %8d  %s" % (name, mode, len(pairs), "bitwise equal" if same else "MISMATCH (%d of %d)" % (int((a.view(np.uint32) != b.view(np.uint32)).sum()), len(a)))
        if reps: msg += "  pipe %.3f ms %.0f GCUPS [%s] | tasks %.3f ms %.0f GCUPS [%s]" % (ta, cells / ta / 1e6, ka, tb, cells / tb / 1e6, kb)
        print(msg, flush=True)
        if not same:
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0][:8]
            print("   first bad pairs:", [(int(p), tuple(int(x) for x in pairs[p]), float(a[p]), float(b[p])) for p in bad], flush=True)
    ar.close()

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
rng = np.random.default_rng(11)
# small, ragged: every code path of the strip / task bookkeeping
for N, mu in ((5, 60), (20, 90), (70, 150), (40, 33)):
    lens = synth_lengths(rng, N, mu)
    profs = [synth_profile(rng, int(L)) for L in lens]
    check("all-pairs N=%d mu=%d" % (N, mu), lens, allpairs.enumerate_pairs(N), profs)
    ordered = np.array([(i, j) for i in range(N) for j in range(N) if i != j], dtype=np.int32)
    check("ordered N=%d mu=%d" % (N, mu), lens, ordered, profs, modes=("global", "local"))
    sub = ordered[rng.random(len(ordered)) < 0.3]
    check("random 30%% N=%d" % N, lens, sub, profs, modes=("global", "semiglobal_both"))
if not quick:
    rng = np.random.default_rng(2); lens = synth_lengths(rng, 256, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    check("C2", lens, allpairs.enumerate_pairs(256), profs, modes=("global", "local", "semiglobal_both"), reps=7)
    rng = np.random.default_rng(4); lens = synth_lengths(rng, 4096, 400)
    pairs = allpairs.enumerate_pairs(4096)
    pairs = pairs[allpairs.shard_columns(lens, pairs, 8)[3]]
    profs = [synth_profile(rng, int(L)) for L in lens]
    check("C4/8", lens, pairs, profs, modes=("global",), reps=3)
