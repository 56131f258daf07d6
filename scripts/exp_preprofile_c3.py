"""build_preprofiles on all of C3 (1024 seqs ~250 aa): stage time from Sequences to ProfileTracks."""
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
for rep in range(3):
    for mode, it in (("local", 2), ("global", 1)):
        t0 = time.perf_counter()
        tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode, waterman_eggert_iterations=it)
        print("build_preprofiles C3 %-6s (%d pass%s, %d alignments with paths): %.1f ms" % (
            mode, it, "es" if it > 1 else "", it * 1024 * 1023, (time.perf_counter() - t0) * 1e3), flush=True)
