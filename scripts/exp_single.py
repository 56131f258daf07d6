"""Latency of ONE alignment with paths (what every TreeMSA merge step pays), raw plan level and through the
PairwiseAligner component."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from praline_amd import native as nat, component as comp, container as ct, core
from praline_amd.matrices import blosum62_matrix
from bench import synth_profile
nat.init(0)
S = blosum62_matrix()
rng = np.random.default_rng(1)
for L in (100, 400, 1000, 3000):
    p1, p2 = synth_profile(rng, L), synth_profile(rng, L)
    for _ in range(2):
        ar = nat.Arena([p1, p2], S); pl = nat.Plan(ar, np.array([(0, 1)], np.int32), want_paths=True)
        pl.run("global", -11, -1); pl.paths(); pl.close(); ar.close()
    t0 = time.perf_counter()
    for _ in range(5):
        ar = nat.Arena([p1, p2], S); pl = nat.Plan(ar, np.array([(0, 1)], np.int32), want_paths=True)
        pl.run("global", -11, -1); sc = pl.scores(); pa = pl.paths(); kms = pl.kernel_ms(); pl.close(); ar.close()
    dt = (time.perf_counter() - t0) / 5
    print("L=%d: arena+plan+run+paths %.2f ms (kernels %.2f ms) -> %.3f GCUPS" % (L, dt * 1e3, kms, L * L / dt / 1e9), flush=True)
print("raw path (praline_build_scores + praline_raw_align, host buffers):")
for L in (100, 400, 1000, 3000):
    p1, p2 = synth_profile(rng, L), synth_profile(rng, L)
    g1 = np.tile(np.array([-11.0, -1.0], np.float32), (L, 1)); g2 = g1.copy()
    for rep in range(3):
        t0 = time.perf_counter()
        m = np.zeros((L, L), np.float32)
        nat.cext_build_scores([p1], [p2], None, None, [S], m)
        t1 = time.perf_counter()
        score, path = nat.raw_align("global", m, g1, g2, None)
        t2 = time.perf_counter()
    print("L=%d: build_scores %.2f ms, raw_align %.2f ms -> %.3f GCUPS" % (L, (t1 - t0) * 1e3, (t2 - t1) * 1e3, L * L / (t2 - t0) / 1e9), flush=True)
