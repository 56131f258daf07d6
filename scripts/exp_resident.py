"""Per-step latency of the progressive alignment (TreeMultipleSequenceAligner) on N sequences ~400 aa: resident
clusters on the GPU (BatchManager) against the host path (serial manager: one PairwiseAligner execution + host
count-track merges + arena re-upload per step)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import component as comp, container as ct, core, native
from bench import synth_lengths

native.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(2)
lens = synth_lengths(rng, N, 400)
base = rng.integers(0, 20, 700)
seqs = []
for i, L in enumerate(lens):
    v = base[:L].copy()
    flip = rng.random(L) < 0.3
    v[flip] = rng.integers(0, 20, int(flip.sum()))
    seqs.append(ct.Sequence("s%03d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]))
idx = core.TypeIndex(); idx.autoregister()
T = [[ct.TRACK_ID_INPUT]]
blosum = ct.blosum62()

def run(manager, cls, keys=None, **inputs):
    ex = core.Execution(manager, "root")
    ex.add_task(cls).environment(core.Environment({}), core.Environment(dict(keys or {}))).inputs(**inputs)
    return core.run(ex)[0]

batch, serial = comp.BatchManager(idx), core.Manager(idx)
t0 = time.perf_counter()
tree = run(batch, comp.GuideTreeBuilder, sequences=seqs, track_id_sets=T, score_matrices=[blosum])['guide_tree']
print("guide tree (%d pairs): %.1f ms" % (N * (N - 1) // 2, (time.perf_counter() - t0) * 1e3))
res = {}
for name, manager in (("resident", batch), ("host", serial), ("resident", batch)):
    t0 = time.perf_counter()
    out = run(manager, comp.TreeMultipleSequenceAligner, sequences=seqs, guide_tree=tree, track_id_sets=T, score_matrices=[blosum])
    dt = time.perf_counter() - t0
    res[name] = np.asarray(out['alignment'].path)
    print("%-8s TreeMSA: %.1f ms total, %.2f ms per merge step (%d steps, final alignment %d columns)" % (
        name, dt * 1e3, dt * 1e3 / (N - 1), N - 1, res[name].shape[0] - 1))
print("equal alignments:", np.array_equal(res["resident"], res["host"]))
lv = comp.merge_levels(list(tree.merge_orders))
print("guide tree levels: %d for %d steps (largest level %d steps)" % (len(lv), N - 1, max(len(x) for x in lv)))
