"""cProfile of the host side of the guide-tree and tree-MSA stages (N = 400)."""
import sys, os, cProfile, pstats, numpy as np
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
N, mu = int(sys.argv[1]) if len(sys.argv) > 1 else 400, 300
rng = np.random.default_rng(7)
anc = rng.integers(0, 20, int(mu * 1.6))
seqs = []
for i, L in enumerate(synth_lengths(rng, N, mu)):
    v = anc[:L].copy(); m = rng.random(L) < 0.35; v[m] = rng.integers(0, 20, int(m.sum()))
    seqs.append(ct.Sequence("s%03d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]))
tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode="global")
pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)), (ct.TRACK_ID_PREPROFILE, t)]) for s, t in zip(seqs, tracks)]
T = [[ct.TRACK_ID_PREPROFILE]]
keys = {"linkage_method": "average", "dist_mode": "global", "merge_mode": "global"}
for stage in ("tree", "msa"):
    pr = cProfile.Profile(); pr.enable()
    if stage == "tree":
        tree = run(comp.GuideTreeBuilder, keys, sequences=pre, track_id_sets=T, score_matrices=[blosum])['guide_tree']
    else:
        msa = run(comp.TreeMultipleSequenceAligner, keys, sequences=pre, guide_tree=tree, track_id_sets=T, score_matrices=[blosum])['alignment']
    pr.disable()
    print("=====", stage)
    pstats.Stats(pr).sort_stats("tottime").print_stats(24)
