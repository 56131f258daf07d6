"""A dumped AdHocMultipleSequenceAligner case of scripts/stress_msa.py (gpurun_out/adhoc_fail.pkl): batching against serial
manager in the default (FAST) and in the reference-order match-score mode."""
import sys, os, pickle, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import native as nat, core, component as comp, container as ct
nat.init(0)
d = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/adhoc_fail.pkl", "rb"))
idx = core.TypeIndex(); idx.autoregister()
batch, serial = comp.BatchManager(idx), core.Manager(idx)
blosum = ct.blosum62()
ss_matrix = ct.ScoreMatrix(None, [ct.ALPHABET_RNA, ct.ALPHABET_RNA], matrix=(np.eye(4, dtype=np.float32) * 3 - 1).astype(np.float32))
seqs = []
for i, tracks in enumerate(d["tracks"]):
    tr = []
    for tid, vals in tracks:
        tr.append((tid, ct.PlainTrack(None, ct.ALPHABET_AA if tid == ct.TRACK_ID_INPUT else ct.ALPHABET_RNA, raw_indices=np.array(vals))))
    seqs.append(ct.Sequence("q%02d" % i, tr))
T = [[ct.TRACK_ID_INPUT], ["ss"]] if d["two_sets"] else [[ct.TRACK_ID_INPUT]]
mats = [blosum, ss_matrix] if d["two_sets"] else [blosum]
for mm_mode in (None, "ref"):
    nat.set_match_mode(mm_mode)
    outs = []
    for manager in (batch, serial):
        ex = core.Execution(manager, "root")
        ex.add_task(comp.AdHocMultipleSequenceAligner).environment(core.Environment({}), core.Environment({"merge_mode": d["mm"], "dist_mode": d["dm"]})).inputs(
            sequences=seqs, track_id_sets=T, score_matrices=mats)
        outs.append(core.run(ex)[0]['alignment'])
    same = [x.name for x in outs[0].items] == [x.name for x in outs[1].items] and np.array_equal(np.asarray(outs[0].path), np.asarray(outs[1].path))
    print("match mode %s: batch == serial: %s (order batch %s / serial %s)" % (mm_mode or "fast", same, [x.name for x in outs[0].items], [x.name for x in outs[1].items]), flush=True)
nat.set_match_mode(None)
