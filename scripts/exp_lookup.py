"""Match-score lookup (k_dp_split16 BSRC = 3, default for one-hot arenas) against the one-hot operand table feeding MFMAs
(PRALINE_NO_LOOKUP=1): bitwise equal scores, kernel times on one rank's share of C4 (one-hot), C2 one-hot and a C5 shard."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, allpairs
from praline_amd.matrices import blosum62_matrix, nucleotide_matrix
from bench import synth_lengths, one_hot
nat.init(0)
def case(tag, lens, profs, S, mine):
    ar = nat.Arena(profs, S)
    cells = int((lens[mine[:, 0]].astype(np.int64) * lens[mine[:, 1]]).sum())
    ref = {}
    for rep in range(2):
        for nl in ("1", "0"):
            os.environ["PRALINE_NO_LOOKUP"] = nl
            pl = nat.Plan(ar, mine)
            for mode in ("global", "local", "semiglobal_both"):
                pl.run(mode, -11, -1)
                ms = []
                for _ in range(4):
                    pl.run(mode, -11, -1); nat.synchronize(); ms.append(pl.kernel_ms())
                sc = pl.scores().copy()
                assert mode not in ref or np.array_equal(ref[mode].view(np.uint32), sc.view(np.uint32)), (tag, mode, "scores differ")
                ref[mode] = sc
                print("%-10s no_lookup=%s %-16s tasks=%d %.2f ms %.0f GCUPS %s" % (tag, nl, mode, pl.tasks, np.median(ms), cells / np.median(ms) / 1e6, pl.kernel_name()), flush=True)
            pl.close()
    ar.close()
rng = np.random.default_rng(4)
N = 4096
lens = synth_lengths(rng, N, 400)
pairs = allpairs.enumerate_pairs(N)
mine = pairs[allpairs.shard_columns(lens, pairs, 8)[3]]
case("C4 share", lens, [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens], blosum62_matrix(), mine)
rng = np.random.default_rng(2)
lens = synth_lengths(rng, 256, 400)
case("C2 onehot", lens, [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens], blosum62_matrix(), np.stack(np.triu_indices(256, 1), axis=1).astype(np.int32))
rng = np.random.default_rng(5)
lens = synth_lengths(rng, 512, 5000)
pairs = allpairs.enumerate_pairs(512)
mine = pairs[allpairs.shard_columns(lens, pairs, 14)[5]]
case("C5 shard", lens, [one_hot(rng.integers(0, 4, int(L)), 15) for L in lens], nucleotide_matrix(), mine)
