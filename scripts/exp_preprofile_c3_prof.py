"""cProfile of build_preprofiles on all of C3 (local, two Waterman-Eggert passes): where the host time goes."""
import sys, os, time, cProfile, pstats, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, component as comp, container as ct
from bench import synth_lengths
nat.init(0)
rng = np.random.default_rng(3)
lens = synth_lengths(rng, 1024, 250)
seqs = [ct.Sequence("s%04d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=rng.integers(0, 20, int(L))))]) for i, L in enumerate(lens)]
blosum = ct.blosum62()
for _ in range(2):
    comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=os.environ.get("PP_MODE", "local"), waterman_eggert_iterations=int(os.environ.get("PP_IT", "2")))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=os.environ.get("PP_MODE", "local"), waterman_eggert_iterations=int(os.environ.get("PP_IT", "2")))
pr.disable()
print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
