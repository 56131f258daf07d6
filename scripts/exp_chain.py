import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat
from praline_amd.matrices import blosum62_matrix
from bench import synth_lengths, synth_profile
nat.init(0)
S = blosum62_matrix()
for N, mu, ordered in ((420, 400, False), (480, 400, False), (560, 400, False), (300, 250, True), (384, 250, True)):
    rng = np.random.default_rng(3)
    lens = synth_lengths(rng, N, mu)
    profs = [synth_profile(rng, int(L)) for L in lens]
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if (i != j if ordered else i < j)], dtype=np.int32)
    cells = int((lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]]).sum())
    ar = nat.Arena(profs, S)
    out = []
    for cm in ("0", "100000000"):
        os.environ["PRALINE_CHAIN_MAX_TASKS"] = cm
        pl = nat.Plan(ar, pairs, want_paths=True)
        pl.run("global", -11, -1); nat.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): pl.run("global", -11, -1)
        nat.synchronize(); dt = (time.perf_counter() - t0) / 3
        out.append("max=%s %.2f ms %.0f GCUPS" % (cm, dt * 1e3, cells / dt / 1e9))
        pl.close()
    print("N=%d mu=%d pairs=%d: %s" % (N, mu, len(pairs), " | ".join(out)), flush=True)
    ar.close()
