"""500 RawPairwiseAligner requests of 400 x 400 as one Execution under BatchManager: wall time of the whole operator path."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, core, component as comp, container as ct
nat.init(0)
idx = core.TypeIndex(); idx.autoregister()
rng = np.random.default_rng(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
def seq(name, n): return ct.Sequence(name, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=np.zeros(n, int)))])
a, b = seq("a", 400), seq("b", 400)
models = []
for _ in range(N):
    m = (rng.standard_normal((400, 400)) * 3 - 0.5).astype(np.float32)
    g = [np.stack([-rng.uniform(5, 12, 400), -rng.uniform(0.5, 2, 400)], axis=1).astype(np.float32) for _ in range(2)]
    models.append((ct.MatchScoreModel(a, b, m), ct.GapScoreModel(a, g[0]), ct.GapScoreModel(b, g[1])))
for name, mgr in (("batch", comp.BatchManager(idx)), ("serial", core.Manager(idx))):
    for rep in range(2):
        t0 = time.perf_counter()
        ex = core.Execution(mgr, "root")
        for mm, g1, g2 in models:
            ex.add_task(comp.RawPairwiseAligner).environment(core.Environment({}), core.Environment({})).inputs(
                mode="global", sequence_one=a, sequence_two=b, match_score_model=mm, gap_score_model_one=g1, gap_score_model_two=g2, zero_idxs=None)
        outs = core.run(ex)
        dt = time.perf_counter() - t0
        print("%s manager: %d requests in %.1f ms (%.2f GCUPS through the operator API)" % (name, N, dt * 1e3, N * 160000 / dt / 1e9), flush=True)
