"""The whole notebook pipeline on the device path: N sequences -> preprofiles (one-call device stage) ->
guide tree (all-pairs scores on the preprofiles) -> TreeMultipleSequenceAligner (N - 1 dependent merges)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, component as comp, container as ct, core
from bench import synth_lengths
nat.init(0)
idx = core.TypeIndex(); idx.autoregister()
manager = comp.BatchManager(idx)
blosum = ct.blosum62()
def run(component, keys=None, **inputs):
    ex = core.Execution(manager, "root")
    ex.add_task(component).environment(core.Environment({}), core.Environment(dict(keys or {}))).inputs(**inputs)
    return core.run(ex)[0]
for N, mu in ((32, 300), (128, 300), (400, 300)):
    rng = np.random.default_rng(7)
    anc = rng.integers(0, 20, int(mu * 1.6))
    seqs = []
    for i, L in enumerate(synth_lengths(rng, N, mu)):
        v = anc[:L].copy(); m = rng.random(L) < 0.35; v[m] = rng.integers(0, 20, int(m.sum()))
        seqs.append(ct.Sequence("s%03d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]))
    t0 = time.perf_counter()
    tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode="global")
    t1 = time.perf_counter()
    pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)), (ct.TRACK_ID_PREPROFILE, t)]) for s, t in zip(seqs, tracks)]
    T = [[ct.TRACK_ID_PREPROFILE]]
    keys = {"linkage_method": "average", "dist_mode": "global", "merge_mode": "global"}
    tree = run(comp.GuideTreeBuilder, keys, sequences=pre, track_id_sets=T, score_matrices=[blosum])['guide_tree']
    t2 = time.perf_counter()
    msa = run(comp.TreeMultipleSequenceAligner, keys, sequences=pre, guide_tree=tree, track_id_sets=T, score_matrices=[blosum])['alignment']
    t3 = time.perf_counter()
    print("N=%d ~%d aa: preprofiles %.0f ms, guide tree %.0f ms, tree MSA (%d merges) %.0f ms (%.1f ms per merge), %d columns" % (
        N, mu, (t1 - t0) * 1e3, (t2 - t1) * 1e3, N - 1, (t3 - t2) * 1e3, (t3 - t2) * 1e3 / (N - 1), np.asarray(msa.path).shape[0] - 1), flush=True)
