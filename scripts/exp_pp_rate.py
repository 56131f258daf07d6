"""build_preprofiles on all of C3 (global one pass / local with two Waterman-Eggert passes): wall time per call, best of 4."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, component as comp, container as ct
from bench import synth_lengths
nat.init(0)
rng = np.random.default_rng(3)
lens = synth_lengths(rng, 1024, 250)
seqs = [ct.Sequence("s%04d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=rng.integers(0, 20, int(L))))]) for i, L in enumerate(lens)]
blosum = ct.blosum62()
for mode, it in (("global", 1), ("local", 2), ("local", 3)):
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode, waterman_eggert_iterations=it)
        best = min(best, time.perf_counter() - t)
    print("build_preprofiles C3 %-6s passes %d: %.1f ms" % (mode, it, best * 1e3), flush=True)
