"""Randomised check of the resident, level-batched TreeMultipleSequenceAligner (BatchManager) against the serial host
path (one PairwiseAligner execution and host merges per step): random sequence sets (one or two track sets), random
guide trees (random merge orders: caterpillars, balanced trees and everything between), every merge mode - equal modes,
scores (1e-5 of max(1, |score|): a profile-profile score near zero is a cancellation of terms of order 100, and the default
match-score mode is good to ~3e-7 of those), paths of every step and equal final alignments.  usage: stress_msa.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from praline_amd import component as comp, container as ct, core, native

native.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
idx = core.TypeIndex(); idx.autoregister()
batch, serial = comp.BatchManager(idx), core.Manager(idx)
blosum = ct.blosum62()
ss_matrix = ct.ScoreMatrix(None, [ct.ALPHABET_RNA, ct.ALPHABET_RNA], matrix=(np.eye(4, dtype=np.float32) * 3 - 1).astype(np.float32))

def run(manager, keys, **inputs):
    ex = core.Execution(manager, "root")
    task = ex.add_task(comp.TreeMultipleSequenceAligner)
    task.environment(core.Environment({}), core.Environment(dict(keys))).inputs(**inputs)
    out = core.run(ex)[0]
    return out, manager.last_instance if hasattr(manager, "last_instance") else None

t_end = time.time() + budget
t_print = time.time()
n_cases = n_steps = n_levels = n_adhoc = n_flips = 0
while time.time() < t_end:
    n = int(rng.choice([2, 3, 5, 9, 17, 33]))
    mu = int(rng.choice([12, 40, 90, 200]))
    two_sets = rng.random() < 0.4
    base = rng.integers(0, 20, 2 * mu)
    seqs = []
    for i in range(n):
        L = int(rng.integers(max(2, mu // 2), mu * 3 // 2 + 1))
        v = base[:L].copy()
        flip = rng.random(L) < rng.choice([0.1, 0.4, 1.0])
        v[flip] = rng.integers(0, 20, int(flip.sum()))
        tracks = [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))]
        if two_sets:
            tracks.append(("ss", ct.PlainTrack(None, ct.ALPHABET_RNA, raw_indices=rng.integers(0, 4, L))))
        seqs.append(ct.Sequence("q%02d" % i, tracks))
    T = [[ct.TRACK_ID_INPUT], ["ss"]] if two_sets else [[ct.TRACK_ID_INPUT]]
    mats = [blosum, ss_matrix] if two_sets else [blosum]
    # a random guide tree: repeatedly merge two random live clusters (j into i)
    alive = list(range(n))
    order = []
    style = rng.choice(["random", "caterpillar", "pairs"])
    while len(alive) > 1:
        if style == "caterpillar":
            a, b = 0, 1
        elif style == "pairs" and len(alive) > 2:
            a = int(rng.integers(0, len(alive) - 1)); b = a + 1
        else:
            a, b = sorted(rng.choice(len(alive), 2, replace=False))
        order.append((alive[a], alive[b]))
        del alive[b]
    tree = ct.SequenceTree(seqs, order)
    merge_mode = str(rng.choice(["semiglobal", "global", "semiglobal_auto"]))
    res = {}
    for name, manager in (("dev", batch), ("host", serial)):
        ex = core.Execution(manager, "root")
        ex.add_task(comp.TreeMultipleSequenceAligner).environment(core.Environment({}), core.Environment({"merge_mode": merge_mode})).inputs(
            sequences=seqs, guide_tree=tree, track_id_sets=T, score_matrices=mats)
        insts = []
        orig = comp.TreeMultipleSequenceAligner.execute
        def spy(self, *a, **k):
            insts.append(self)
            return orig(self, *a, **k)
        comp.TreeMultipleSequenceAligner.execute = spy
        try:
            out = core.run(ex)[0]
        finally:
            comp.TreeMultipleSequenceAligner.execute = orig
        res[name] = (insts[0], out)
    dev, host = res["dev"][0], res["host"][0]
    ok = len(dev.steps) == len(host.steps) == n - 1
    for (m1, s1, p1), (m2, s2, p2) in zip(host.steps, dev.steps):
        ok = ok and m1 == m2 and abs(s1 - s2) <= 1e-5 * max(1.0, abs(s1)) and np.array_equal(p1, p2)
    ok = ok and np.array_equal(np.asarray(res["dev"][1]['alignment'].path), np.asarray(res["host"][1]['alignment'].path))
    if not ok:
        print("MISMATCH n=%d mu=%d two_sets=%s style=%s merge_mode=%s order=%s" % (n, mu, two_sets, style, merge_mode, order), flush=True)
        for c, ((m1, s1, p1), (m2, s2, p2)) in enumerate(zip(host.steps, dev.steps)):
            if not (m1 == m2 and abs(s1 - s2) <= 1e-5 * max(1.0, abs(s1)) and np.array_equal(p1, p2)):
                print("  first differing step %d %s: host %s %r rows %d / device %s %r rows %d" % (c, order[c], m1, s1, len(p1), m2, s2, len(p2)), flush=True)
                print("  host path   %s" % np.asarray(p1).tolist(), flush=True)
                print("  device path %s" % np.asarray(p2).tolist(), flush=True)
                break
        print("  device levels %s" % dev.levels, flush=True)
        print("  sequences %s" % [s_.get_track(ct.TRACK_ID_INPUT).values.tolist() for s_ in seqs], flush=True)
        sys.exit(1)
    n_cases += 1; n_steps += n - 1; n_levels += len(dev.levels)
    if n <= 17 and rng.random() < 0.3:
        # AdHocMultipleSequenceAligner (msa.py:250-558): resident clusters + score cache under the batching manager against
        # the serial manager - equal final alignments
        mm, dm = [("semiglobal", "global"), ("global", "global"), ("semiglobal_auto", "semiglobal_auto"), ("global", "semiglobal")][int(rng.integers(0, 4))]
        outs = []
        def adhoc_pair():
            got = []
            for manager in (batch, serial):
                ex = core.Execution(manager, "root")
                ex.add_task(comp.AdHocMultipleSequenceAligner).environment(core.Environment({}), core.Environment({"merge_mode": mm, "dist_mode": dm})).inputs(
                    sequences=seqs, track_id_sets=T, score_matrices=mats)
                got.append(core.run(ex)[0]['alignment'])
            return got
        def differ(o):
            return [x.name for x in o[0].items] != [x.name for x in o[1].items] or not np.array_equal(np.asarray(o[0].path), np.asarray(o[1].path))
        outs = adhoc_pair()
        if differ(outs):
            # The batching manager scores cluster pairs with scores-only plans (f16 hi/lo split), the serial one with single
            # alignments (fp32 chain): on merged float profiles the two agree to ~3e-7, and two cluster pairs whose scores tie
            # to within that can be joined in a different order (1 case in ~4 700).  Not a defect as long as the
            # reference-order mode - bit-identical scores on every path - gives one answer: checked here.
            native.set_match_mode("ref")
            try:
                again = adhoc_pair()
            finally:
                native.set_match_mode(None)
            if not differ(again):
                n_flips += 1
                n_adhoc += 1
                continue
        if differ(outs):
            print("ADHOC MISMATCH n=%d mu=%d two_sets=%s merge_mode=%s dist_mode=%s" % (n, mu, two_sets, mm, dm), flush=True)
            import pickle
            pickle.dump({"tracks": [[(tid, s_.get_track(tid).values.tolist()) for tid in ([ct.TRACK_ID_INPUT, "ss"] if two_sets else [ct.TRACK_ID_INPUT])] for s_ in seqs],
                         "two_sets": two_sets, "mm": mm, "dm": dm, "paths": [np.asarray(o.path).tolist() for o in outs],
                         "names": [[x.name for x in o.items] for o in outs]}, open("gpurun_out/adhoc_fail.pkl", "wb"))
            print("  sequences %s" % [s_.get_track(ct.TRACK_ID_INPUT).values.tolist() for s_ in seqs], flush=True)
            sys.exit(1)
        n_adhoc += 1
    if time.time() - t_print > 60:
        t_print = time.time()
        print("  ... %d alignments, %d merge steps in %d device levels" % (n_cases, n_steps, n_levels), flush=True)
print("stress_msa ok: %d progressive alignments, %d merge steps in %d device levels, all equal to the serial host path; %d ad-hoc alignments equal (%d of them only in the reference-order mode: a join order decided by a 1e-7 score difference)" % (n_cases, n_steps, n_levels, n_adhoc, n_flips))
