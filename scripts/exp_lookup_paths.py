"""Match-score lookup in the path kernels of one-hot plans (k_dp_split16_tb BSRC = 3: chain mode and task mode, integer
scoring) against the MFMA instances (PRALINE_NO_LOOKUP=1): identical scores and paths, rates on C2 one-hot (chain mode)
and on all of C3 (task mode; local runs the two-pass scheme whose kernels are not affected)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, one_hot
nat.init(0)
S = blosum62_matrix()
def case(tag, lens, profs, pairs, modes, reps=4):
    ar = nat.Arena(profs, S)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ref = {}
    for rep in range(2):
        for nl in ("1", "0"):
            os.environ["PRALINE_NO_LOOKUP"] = nl
            pl = nat.Plan(ar, pairs, want_paths=True)
            for mode in modes:
                pl.run(mode, -11, -1); nat.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    pl.run(mode, -11, -1)
                nat.synchronize()
                dt = (time.perf_counter() - t) / reps
                sc = pl.scores().copy(); buf, o, r = pl.paths_packed()
                if mode in ref:
                    assert np.array_equal(ref[mode][0].view(np.uint32), sc.view(np.uint32)), (tag, mode, "scores")
                    assert np.array_equal(ref[mode][2], r), (tag, mode, "path lengths")
                    sel = np.random.default_rng(1).permutation(len(r))[:20000]
                    for q in sel:
                        assert np.array_equal(ref[mode][1][ref[mode][3][q]:ref[mode][3][q] + r[q]], buf[o[q]:o[q] + r[q]]), (tag, mode, q)
                else:
                    ref[mode] = (sc, buf.copy(), r.copy(), o.copy())
                print("%-10s no_lookup=%s %-16s %.2f ms %.0f GCUPS %s" % (tag, nl, mode, dt * 1e3, cells / dt / 1e9, pl.kernel_name()), flush=True)
            pl.close()
    ar.close()
rng = np.random.default_rng(2)
lens = synth_lengths(rng, 256, 400)
case("C2 onehot", lens, [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens], np.stack(np.triu_indices(256, 1), axis=1).astype(np.int32),
     ("global", "local", "semiglobal_both"))
rng = np.random.default_rng(3)
lens = synth_lengths(rng, 1024, 250)
i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
case("C3", lens, [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens], p3, ("global", "semiglobal_both", "local"), reps=2)
