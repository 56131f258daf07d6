"""Whole preprofile stage on the device (component.build_preprofiles): N sequences ~250 aa, every sequence
as master against all others (N(N-1) alignments with paths per pass), counts only come back."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, component as comp, container as ct
from bench import synth_lengths
nat.init(0)
blosum = ct.blosum62()
for N in (128, 512):
    rng = np.random.default_rng(3)
    lens = synth_lengths(rng, N, 250)
    seqs = [ct.Sequence("s%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=rng.integers(0, 20, int(L))))])
            for i, L in enumerate(lens)]
    cells = int(lens.sum()) ** 2 - int((lens.astype(np.int64) ** 2).sum())
    for mode, passes in (("global", 1), ("local", 2)):
        comp.build_preprofiles(seqs[:8], ct.TRACK_ID_INPUT, blosum, mode=mode)
        t0 = time.perf_counter()
        tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode)
        dt = time.perf_counter() - t0
        print("N=%d %-6s %d pass(es): %.0f ms total (host + device), %.0f GCUPS end to end; counts sum %d" % (
            N, mode, passes, dt * 1e3, cells * passes / dt / 1e9, sum(int(t.counts.sum()) for t in tracks)), flush=True)
