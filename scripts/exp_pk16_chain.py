"""k_dp_pk16_tb in chain mode on C2-sized one-hot plans: the publish interval (PRALINE_CHAIN_EVERY, rows) against the rate;
a single long alignment and a merge-step sized batch beside it."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat
from bench import make_workload, one_hot, synth_lengths
nat.init(0)
w = make_workload("c2")
rng = np.random.default_rng(2)

def rate(tag, lens, pairs, modes=("global", "local")):
    oh = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
    arena = nat.Arena(oh, w["S"])
    cells = int((np.asarray(lens)[pairs[:, 0]].astype(np.int64) * np.asarray(lens)[pairs[:, 1]]).sum())
    for mode in modes:
        out = []
        for every in os.environ.get("EVERY", "0,6,12,24,48,96").split(","):
            if every == "0":
                os.environ.pop("PRALINE_CHAIN_EVERY", None)
            else:
                os.environ["PRALINE_CHAIN_EVERY"] = every
            for pk in os.environ.get("PK", "1").split(","):
                os.environ["PRALINE_TB_PK16"] = pk
                plan = nat.Plan(arena, pairs, want_paths=True)
                for _ in range(2):
                    plan.run(mode, -11.0, -1.0)
                nat.synchronize()
                best = 1e9
                for _ in range(4):
                    t = time.perf_counter(); plan.run(mode, -11.0, -1.0); nat.synchronize(); best = min(best, time.perf_counter() - t)
                out.append("%s/%s: %.3f ms %.0f" % (every, pk, best * 1e3, cells / best / 1e9))
                name = plan.kernel_name()
                plan.close()
        print("%-26s %-8s %s   [%s]" % (tag, mode, " | ".join(out), name), flush=True)
    arena.close()

n = len(w["lens"])
rate("C2 one-hot 32640 pairs", w["lens"], np.stack(np.triu_indices(n, 1), axis=1).astype(np.int32))
l8 = synth_lengths(rng, 64, 400)
rate("64 x ~400, 2016 pairs", l8, np.stack(np.triu_indices(64, 1), axis=1).astype(np.int32))
rate("one pair 400 x 400", [400, 400], np.array([[0, 1]], dtype=np.int32), modes=("global",))
rate("one pair 1400 x 1400", [1400, 1400], np.array([[0, 1]], dtype=np.int32), modes=("global",))
rate("128 x ~1000, 8128 pairs", synth_lengths(rng, 128, 1000), np.stack(np.triu_indices(128, 1), axis=1).astype(np.int32), modes=("global",))
