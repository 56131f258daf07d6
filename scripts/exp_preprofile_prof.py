"""cProfile of component.build_preprofiles (N = 512, global and local): where the host time of the device
preprofile stage goes."""
import sys, os, time, cProfile, pstats, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, component as comp, container as ct
from bench import synth_lengths
nat.init(0)
blosum = ct.blosum62()
N = int(os.environ.get("N", "512"))
rng = np.random.default_rng(3)
lens = synth_lengths(rng, N, 250)
seqs = [ct.Sequence("s%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=rng.integers(0, 20, int(L))))])
        for i, L in enumerate(lens)]
comp.build_preprofiles(seqs[:8], ct.TRACK_ID_INPUT, blosum, mode="global")
for mode in ("global", "local"):
    comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode)
    dt = time.perf_counter() - t0
    pr.disable()
    print("=====", mode, "%.1f ms" % (dt * 1e3))
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
