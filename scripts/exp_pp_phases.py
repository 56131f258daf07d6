"""Phases of the preprofile stage on C3 (global, one pass), each fenced with a device synchronisation."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native, component as comp, container as ct
from bench import synth_lengths, one_hot
native.init(0)
rng = np.random.default_rng(3)
lens = synth_lengths(rng, 1024, 250)
profs = [one_hot(rng.integers(0, 20, int(L)), 27) for L in lens]
S = np.asarray(ct.blosum62().matrix, dtype=np.float32)
n = 1024
pairs = np.empty((n * (n - 1), 2), dtype=np.int32)
pairs[:, 0] = np.repeat(np.arange(n, dtype=np.int32), n - 1)
sl = np.tile(np.arange(n - 1, dtype=np.int32), n)
pairs[:, 1] = sl + (sl >= pairs[:, 0])
for rep in range(3):
    T = [time.perf_counter()]
    def mark(): native.synchronize(); T.append(time.perf_counter())
    arena = native.Arena(profs, S); arena.counts_reset(); mark()
    plan = native.Plan(arena, pairs, want_paths=True); mark()
    plan.run("global", -11.0, -1.0); mark()
    plan.add_counts(None, local=False); mark()
    plan.close(); mark()
    c = arena.counts(); mark()
    arena.close(); mark()
    names = ["arena", "plan", "run", "add_counts", "plan.close", "counts D2H", "arena.close"]
    print("  ".join("%s %.1f" % (nm, (T[i + 1] - T[i]) * 1e3) for i, nm in enumerate(names)), " total %.1f ms" % ((T[-1] - T[0]) * 1e3), flush=True)
