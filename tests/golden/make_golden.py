#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REAL REFERENCE (ibivu/PRALINE) in this container.

The reference package is imported in place from /root/reference (oracle/ref_import.py), with its
C extension compiled from the reference's own cext.c by oracle/build_ref.sh.  The outputs are
plain data (inputs + expected outputs); no reference source is stored.  Run:

    python3 tests/golden/make_golden.py

Fixtures (all arrays little-endian numpy):
  bba0184_inputs.npz       the 5 sequences of extra/data/BBA0184.tfa as index arrays, BLOSUM62
                           as the 27x27 float32 matrix the reference loader produces, motif/ss
                           tracks and their matrices
  kat_pairwise.npz         PairwiseAligner score + path for all 10 pairs x 5 modes
  fill_small.npz           captured native calls (m, g1, g2, o/t before and after, z) for small
                           cases in all modes incl. zero_idxs masks and float profile scoring
  preprofile.npz           Global/LocalMasterSlaveAligner (Waterman-Eggert) + ProfileBuilder:
                           every inner PairwiseAligner call (zero_idxs, score, path), merged
                           master-slave alignment paths, profile counts
  profile_profile.npz      PairwiseAligner on preprofile tracks (float scoring): profiles, m,
                           score, path for all pairs; GuideTreeBuilder distance matrix
  treemsa.npz              GuideTreeBuilder + TreeMultipleSequenceAligner on preprofile and input tracks,
                           merge modes semiglobal / global / semiglobal_auto: every merge step's mode,
                           score and path, the final alignment path
  adhoc.npz                AdHocMultipleSequenceAligner: the sequence of inner PairwiseAligner calls (mode,
                           cluster names, lengths, score) and the final alignment, 4 mode combinations
  multitrack.npz           num_sets = 2 and 3 match-score matrices + alignments
  synthetic_c1.npz         BASELINE config 0: seed 1, N=8, mu=100, one-hot, BLOSUM62, global
  synthetic_dna.npz        small DNA (A=15, packaged nucleotide matrix) cases, all modes
  matrices.npz             every packaged score table (praline/matrices/*) as the reference's load_score_matrix
                           parses it: float32 [27, 27] over ALPHABET_AA (the BLOSUM tables) / [15, 15] over ALPHABET_DNA
  BBA0184.{cli,multitrack}.aln, BBA0184.{motif,ss}.tfa, motif_score_matrix
                           the reference's shipped known-answer alignments and their annotation inputs (extra/data),
                           byte copies - data the end-to-end tests reproduce   (`make_golden.py matrices` writes only
                           matrices.npz and these copies)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference, REF_ROOT  # noqa: E402

praline = import_reference()
import praline.component as pc  # noqa: E402
import praline.component.align as pc_align  # noqa: E402
import praline.container as ct  # noqa: E402
import praline.core as core  # noqa: E402
from praline import load_score_matrix, load_sequence_fasta, open_builtin  # noqa: E402
from praline.util import get_frequencies  # noqa: E402

MODES = ["global", "local", "semiglobal_both", "semiglobal_one", "semiglobal_two"]
DATA = os.path.join(REF_ROOT, "extra", "data")


def make_manager():
    idx = core.TypeIndex()
    for name in dir(pc):
        c = getattr(pc, name)
        if isinstance(c, type) and issubclass(c, core.Component) and c is not core.Component:
            idx.register(c)
    return core.Manager(idx)


MAN = make_manager()


def run(component, env_keys=None, **inputs):
    ex = core.Execution(MAN, "root")
    t = ex.add_task(component)
    t.environment(core.Environment({}), core.Environment(dict(env_keys or {})))
    t.inputs(**inputs)
    for _ in ex.run():
        pass
    return ex.outputs[0]


def pairwise(mode, s1, s2, tracks, sms, zero_idxs=None, env=None):
    out = run(pc.PairwiseAligner, env, mode=mode, sequence_one=s1, sequence_two=s2,
              track_id_sets_one=tracks, track_id_sets_two=tracks, score_matrices=sms,
              zero_idxs=zero_idxs)
    return float(out["score"]), np.array(out["alignment"].path, dtype=np.int64)


class Capture(object):
    """Records every native call the reference makes (build_scores + fill)."""

    def __init__(self):
        self.fills = []
        self.builds = []
        self._orig_align = dict(pc_align._CEXT_ALIGN_FUNCTIONS)
        self._orig_build = pc_align.cext_build_scores

    def __enter__(self):
        def wrap(mode, fn):
            def inner(m, g1, g2, o, t, z):
                rec = dict(mode=mode, m=m.copy(), g1=g1.copy(), g2=g2.copy(), o0=o.copy(),
                           t0=t.copy(), z=z.copy())
                fn(m, g1, g2, o, t, z)
                rec["o"] = o.copy()
                rec["t"] = t.copy()
                self.fills.append(rec)
            return inner
        for mode, fn in self._orig_align.items():
            pc_align._CEXT_ALIGN_FUNCTIONS[mode] = wrap(mode, fn)

        def build(i1, i2, i1nz, i2nz, s, m):
            self._orig_build(i1, i2, i1nz, i2nz, s, m)
            self.builds.append(dict(i1=[a.copy() for a in i1], i2=[a.copy() for a in i2],
                                    s=[a.copy() for a in s], m=m.copy()))
        pc_align.cext_build_scores = build
        return self

    def __exit__(self, *a):
        pc_align._CEXT_ALIGN_FUNCTIONS.update(self._orig_align)
        pc_align.cext_build_scores = self._orig_build


def pack_paths(paths):
    off = np.zeros(len(paths) + 1, dtype=np.int64)
    for i, p in enumerate(paths):
        off[i + 1] = off[i] + p.shape[0]
    width = paths[0].shape[1] if paths else 2
    cat = np.concatenate(paths, axis=0) if paths else np.zeros((0, width), np.int64)
    return cat.astype(np.int64), off


def sha8(path):
    return hashlib.sha1(np.array(path, dtype=np.int64).tobytes()).hexdigest()[:8]


def save(name, **arrays):
    fn = os.path.join(HERE, name)
    np.savez_compressed(fn, **arrays)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(fn) / 1024.0))


def main():
    aa = ct.ALPHABET_AA
    blosum62 = load_score_matrix(open_builtin("matrices/blosum62"), alphabet=aa)
    seqs = load_sequence_fasta(os.path.join(DATA, "BBA0184.tfa"), aa)
    T_IN = [[ct.TRACK_ID_INPUT]]

    # ---------------------------------------------------------------- inputs
    motif_alpha = ct.Alphabet("golden.motif", [("*", 0), ("M", 1)])
    ss_alpha = ct.Alphabet("golden.ss", [("C", 0), ("H", 1), ("E", 2)])
    motif_seqs = load_sequence_fasta(os.path.join(DATA, "BBA0184.motif.tfa"), motif_alpha)
    ss_seqs = load_sequence_fasta(os.path.join(DATA, "BBA0184.ss.tfa"), ss_alpha)
    motif_sm = load_score_matrix(os.path.join(DATA, "motif_score_matrix"), alphabet=motif_alpha)
    ss_scores = {}
    for a in "CHE":
        for b in "CHE":
            ss_scores[(a, b)] = 3.0 if a == b else 0.0
    ss_sm = ct.ScoreMatrix(ss_scores, [ss_alpha, ss_alpha])
    inp = dict(blosum62=blosum62.matrix.astype(np.float32),
               motif_matrix=motif_sm.matrix.astype(np.float32),
               ss_matrix=ss_sm.matrix.astype(np.float32),
               names=np.array([s.name for s in seqs]))
    for i, s in enumerate(seqs):
        inp["seq%d" % i] = s.get_track(ct.TRACK_ID_INPUT).values.astype(np.int32)
        inp["motif%d" % i] = motif_seqs[i].get_track(ct.TRACK_ID_INPUT).values.astype(np.int32)
        inp["ss%d" % i] = ss_seqs[i].get_track(ct.TRACK_ID_INPUT).values.astype(np.int32)
    save("bba0184_inputs.npz", **inp)

    # ---------------------------------------------------------------- KATs (Appendix B)
    kat = {}
    rows = []
    for i in range(5):
        for j in range(i + 1, 5):
            for mode in MODES:
                sc, path = pairwise(mode, seqs[i], seqs[j], T_IN, [blosum62])
                kat["path_%d_%d_%s" % (i, j, mode)] = path
                kat["score_%d_%d_%s" % (i, j, mode)] = np.float64(sc)
                rows.append("(%d,%d) %-16s %8.1f %4d %s" % (i, j, mode, sc, path.shape[0], sha8(path)))
    save("kat_pairwise.npz", **kat)
    with open(os.path.join(HERE, "kat_pairwise.txt"), "w") as f:
        f.write("# pair mode score rows sha1(int64 path)[:8] - from the real reference\n")
        f.write("\n".join(rows) + "\n")

    # ---------------------------------------------------------------- captured native calls
    rng = np.random.default_rng(12345)
    fills = {}
    n = 0

    def small_seq(name, L, alphabet=aa, hi=20):
        vals = rng.integers(0, hi, L)
        tr = ct.PlainTrack(None, alphabet, raw_indices=vals)
        return ct.Sequence(name, [(ct.TRACK_ID_INPUT, tr)])

    def small_prof(name, L, alphabet=aa, hi=20):
        counts = np.zeros((L, alphabet.size), dtype=int)
        for r in range(L):
            k = rng.integers(1, 8)
            idx = rng.choice(hi, size=k, replace=False)
            counts[r, idx] = rng.integers(1, 6, k)
        tr = ct.ProfileTrack(counts, alphabet)
        return ct.Sequence(name, [(ct.TRACK_ID_INPUT, tr)])

    cases = []
    for (L1, L2) in [(1, 1), (1, 7), (9, 1), (13, 17), (40, 33), (64, 65), (31, 96)]:
        cases.append(("onehot", small_seq("a", L1), small_seq("b", L2), None, [-11.0, -1.0]))
    cases.append(("onehot_linear", small_seq("a", 25), small_seq("b", 30), None, [-4.0]))
    cases.append(("onehot_gap0", small_seq("a", 20), small_seq("b", 22), None, [-3.0, -3.0]))
    cases.append(("profile", small_prof("a", 37), small_prof("b", 41), None, [-11.0, -1.0]))
    cases.append(("profile", small_prof("a", 5), small_prof("b", 66), None, [-6.5, -0.75]))
    cases.append(("mixed", small_seq("a", 30), small_prof("b", 28), None, [-11.0, -1.0]))
    cases.append(("mask", small_seq("a", 40), small_seq("b", 45),
                  [(y, x) for y in range(10, 21) for x in range(12, 30)], [-11.0, -1.0]))
    cases.append(("mask_scatter", small_seq("a", 24), small_seq("b", 24),
                  [(int(rng.integers(1, 25)), int(rng.integers(1, 25))) for _ in range(40)],
                  [-11.0, -1.0]))
    for kind, s1, s2, zi, gaps in cases:
        for mode in MODES:
            with Capture() as cap:
                sc, path = pairwise(mode, s1, s2, T_IN, [blosum62], zero_idxs=zi,
                                    env={"gap_series": gaps})
            rec, b = cap.fills[0], cap.builds[0]
            p = "c%03d_" % n
            fills[p + "mode"] = np.array(mode)
            fills[p + "kind"] = np.array(kind)
            fills[p + "gaps"] = np.array(gaps, dtype=np.float64)
            fills[p + "p1"] = b["i1"][0]
            fills[p + "p2"] = b["i2"][0]
            fills[p + "m"] = rec["m"]
            fills[p + "g1"] = rec["g1"]
            fills[p + "g2"] = rec["g2"]
            fills[p + "z"] = rec["z"]
            fills[p + "o0"] = rec["o0"]
            fills[p + "t0"] = rec["t0"]
            fills[p + "o"] = rec["o"]
            fills[p + "t"] = rec["t"]
            fills[p + "score"] = np.float64(sc)
            fills[p + "path"] = path
            if zi is not None:
                fills[p + "zero_idxs"] = np.array(zi, dtype=np.int64)
            n += 1
    fills["n_cases"] = np.int64(n)
    save("fill_small.npz", **fills)

    # ---------------------------------------------------------------- preprofile stage
    pre = {}
    for kind, comp, env in (("global", pc.GlobalMasterSlaveAligner, {}),
                            ("local", pc.LocalMasterSlaveAligner, {}),
                            ("local_thr", pc.LocalMasterSlaveAligner, {"score_threshold": 100.0}),
                            ("local_we3", pc.LocalMasterSlaveAligner,
                             {"waterman_eggert_iterations": 3})):
        for mi in ([0, 4] if kind != "global" else [0, 2, 4]):
            master = seqs[mi]
            slaves = [s for k, s in enumerate(seqs) if k != mi]
            calls = []
            orig = pc.PairwiseAligner.execute

            def spy(self, mode, sequence_one, sequence_two, track_id_sets_one,
                    track_id_sets_two, zero_idxs, score_matrices, _orig=orig, _calls=calls):
                zi = None if zero_idxs is None else list(zero_idxs)
                for msg in _orig(self, mode, sequence_one, sequence_two, track_id_sets_one,
                                 track_id_sets_two, zero_idxs, score_matrices):
                    if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.outputs is not None \
                            and "alignment" in msg.outputs:
                        _calls.append((mode, sequence_two.name, zi, float(msg.outputs["score"]),
                                       np.array(msg.outputs["alignment"].path, dtype=np.int64)))
                    yield msg
            pc.PairwiseAligner.execute = spy
            try:
                out = run(comp, env, master_sequence=master, slave_sequences=slaves,
                          track_id_sets=T_IN, score_matrices=[blosum62])
            finally:
                pc.PairwiseAligner.execute = orig
            aln = out["alignment"]
            key = "%s_m%d_" % (kind, mi)
            pre[key + "msa_path"] = np.array(aln.path, dtype=np.int64)
            pre[key + "msa_names"] = np.array([s.name for s in aln.items])
            pre[key + "n_calls"] = np.int64(len(calls))
            for c, (mode, sname, zi, sc, path) in enumerate(calls):
                pre[key + "call%d_mode" % c] = np.array(mode)
                pre[key + "call%d_slave" % c] = np.array(sname)
                pre[key + "call%d_score" % c] = np.float64(sc)
                pre[key + "call%d_path" % c] = path
                # zero_idxs are full rectangles (preprofile.py:247-255): store their corners
                if zi:
                    a = np.array(zi, dtype=np.int64)
                    pre[key + "call%d_zero_count" % c] = np.int64(a.shape[0])
                    pre[key + "call%d_zero_sha" % c] = np.array(
                        hashlib.sha1(a.tobytes()).hexdigest())
            prof = run(pc.ProfileBuilder, {}, alignment=aln, track_id=ct.TRACK_ID_INPUT)
            pre[key + "profile_counts"] = np.array(prof["profile_track"].counts, dtype=np.int64)
            pre[key + "profile_f32"] = prof["profile_track"].profile.astype(np.float32)
    save("preprofile.npz", **pre)

    # ---------------------------------------------------------------- profile-profile (float)
    pp = {}
    pre_seqs = []
    for mi in range(5):
        master = seqs[mi]
        slaves = [s for k, s in enumerate(seqs) if k != mi]
        out = run(pc.GlobalMasterSlaveAligner, {}, master_sequence=master, slave_sequences=slaves,
                  track_id_sets=T_IN, score_matrices=[blosum62])
        prof = run(pc.ProfileBuilder, {}, alignment=out["alignment"], track_id=ct.TRACK_ID_INPUT)
        s = ct.Sequence(master.name, [(ct.TRACK_ID_INPUT, master.get_track(ct.TRACK_ID_INPUT)),
                                      (ct.TRACK_ID_PREPROFILE, prof["profile_track"])])
        pre_seqs.append(s)
        pp["counts%d" % mi] = np.array(prof["profile_track"].counts, dtype=np.int64)
        pp["profile%d" % mi] = prof["profile_track"].profile.astype(np.float32)
    T_PRE = [[ct.TRACK_ID_PREPROFILE]]
    d = np.zeros((5, 5), dtype=np.float32)
    for i in range(5):
        for j in range(i + 1, 5):
            for mode in MODES:
                with Capture() as cap:
                    sc, path = pairwise(mode, pre_seqs[i], pre_seqs[j], T_PRE, [blosum62])
                pp["score_%d_%d_%s" % (i, j, mode)] = np.float64(sc)
                pp["path_%d_%d_%s" % (i, j, mode)] = path
                if mode == "global":
                    d[i, j] = d[j, i] = sc
                    if (i, j) in ((0, 4), (2, 3)):
                        pp["m_%d_%d" % (i, j)] = cap.builds[0]["m"]
    pp["d"] = d
    pp["dist"] = (-d) + d.max()  # tree.py:147
    tree = run(pc.GuideTreeBuilder, {}, sequences=pre_seqs, track_id_sets=T_PRE,
               score_matrices=[blosum62])
    pp["merge_order"] = np.array(list(tree["guide_tree"].merge_orders), dtype=np.int64)
    save("profile_profile.npz", **pp)

    # ---------------------------------------------------------------- progressive MSA (msa.py:124-237)
    msa = {}
    tree_in = run(pc.GuideTreeBuilder, {}, sequences=seqs, track_id_sets=T_IN, score_matrices=[blosum62])
    msa["merge_order_input"] = np.array(list(tree_in["guide_tree"].merge_orders), dtype=np.int64)
    for tag, sset, tracks, gtree in (("pre", pre_seqs, T_PRE, tree["guide_tree"]),
                                     ("in", seqs, T_IN, tree_in["guide_tree"])):
        for merge_mode in ("semiglobal", "global", "semiglobal_auto"):
            calls = []
            orig = pc.PairwiseAligner.execute

            def spy(self, mode, sequence_one, sequence_two, track_id_sets_one, track_id_sets_two,
                    zero_idxs, score_matrices, _orig=orig, _calls=calls):
                for msg in _orig(self, mode, sequence_one, sequence_two, track_id_sets_one,
                                 track_id_sets_two, zero_idxs, score_matrices):
                    if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.outputs is not None \
                            and "alignment" in msg.outputs:
                        _calls.append((mode, float(msg.outputs["score"]),
                                       np.array(msg.outputs["alignment"].path, dtype=np.int64),
                                       len(sequence_one), len(sequence_two)))
                    yield msg
            pc.PairwiseAligner.execute = spy
            try:
                out = run(pc.TreeMultipleSequenceAligner, {"merge_mode": merge_mode}, sequences=sset,
                          guide_tree=gtree, track_id_sets=tracks, score_matrices=[blosum62])
            finally:
                pc.PairwiseAligner.execute = orig
            key = "%s_%s_" % (tag, merge_mode)
            msa[key + "path"] = np.array(out["alignment"].path, dtype=np.int64)
            msa[key + "names"] = np.array([s.name for s in out["alignment"].items])
            msa[key + "n_steps"] = np.int64(len(calls))
            for c, (mode, sc, path, l1, l2) in enumerate(calls):
                msa[key + "step%d_mode" % c] = np.array(mode)
                msa[key + "step%d_score" % c] = np.float64(sc)
                msa[key + "step%d_path" % c] = path
                msa[key + "step%d_lens" % c] = np.array([l1, l2], dtype=np.int64)
    save("treemsa.npz", **msa)

    # ---------------------------------------------------------------- ad hoc MSA (msa.py:250-558)
    adhoc = {}
    for tag, sset, tracks in (("pre", pre_seqs, T_PRE), ("in", seqs, T_IN)):
        for merge_mode, dist_mode in (("semiglobal", "global"), ("global", "global"),
                                      ("semiglobal_auto", "semiglobal_auto"), ("global", "semiglobal")):
            calls = []
            orig = pc.PairwiseAligner.execute

            def spy(self, mode, sequence_one, sequence_two, track_id_sets_one, track_id_sets_two,
                    zero_idxs, score_matrices, _orig=orig, _calls=calls):
                for msg in _orig(self, mode, sequence_one, sequence_two, track_id_sets_one,
                                 track_id_sets_two, zero_idxs, score_matrices):
                    if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.outputs is not None \
                            and "alignment" in msg.outputs:
                        _calls.append((mode, sequence_one.name, sequence_two.name, len(sequence_one),
                                       len(sequence_two), float(msg.outputs["score"])))
                    yield msg
            pc.PairwiseAligner.execute = spy
            try:
                out = run(pc.AdHocMultipleSequenceAligner, {"merge_mode": merge_mode, "dist_mode": dist_mode},
                          sequences=sset, track_id_sets=tracks, score_matrices=[blosum62])
            finally:
                pc.PairwiseAligner.execute = orig
            key = "%s_%s_%s_" % (tag, merge_mode, dist_mode)
            adhoc[key + "path"] = np.array(out["alignment"].path, dtype=np.int64)
            adhoc[key + "names"] = np.array([s.name for s in out["alignment"].items])
            adhoc[key + "call_modes"] = np.array([c[0] for c in calls])
            adhoc[key + "call_one"] = np.array([c[1] for c in calls])
            adhoc[key + "call_two"] = np.array([c[2] for c in calls])
            adhoc[key + "call_lens"] = np.array([[c[3], c[4]] for c in calls], dtype=np.int64)
            adhoc[key + "call_scores"] = np.array([c[5] for c in calls], dtype=np.float64)
    save("adhoc.npz", **adhoc)

    # ---------------------------------------------------------------- multi-track sets
    mt = {}
    mseqs = []
    for i in range(5):
        tr = [(ct.TRACK_ID_INPUT, seqs[i].get_track(ct.TRACK_ID_INPUT)),
              ("golden.motif", motif_seqs[i].get_track(ct.TRACK_ID_INPUT)),
              ("golden.ss", ss_seqs[i].get_track(ct.TRACK_ID_INPUT))]
        mseqs.append(ct.Sequence(seqs[i].name, tr))
    for nsets, tracks, sms in ((2, [[ct.TRACK_ID_INPUT], ["golden.motif"]], [blosum62, motif_sm]),
                               (3, [[ct.TRACK_ID_INPUT], ["golden.motif"], ["golden.ss"]],
                                [blosum62, motif_sm, ss_sm])):
        for (i, j) in ((0, 1), (0, 4), (2, 3)):
            for mode in ("global", "local", "semiglobal_both"):
                with Capture() as cap:
                    sc, path = pairwise(mode, mseqs[i], mseqs[j], tracks, sms)
                k = "s%d_%d_%d_%s_" % (nsets, i, j, mode)
                mt[k + "score"] = np.float64(sc)
                mt[k + "path"] = path
                if mode == "global" and (i, j) == (0, 4):
                    mt[k + "m"] = cap.builds[0]["m"]
    save("multitrack.npz", **mt)

    # ---------------------------------------------------------------- synthetic C1
    def synth(seed, N, mu, hi):
        r = np.random.default_rng(seed)
        lens = np.clip(np.rint(r.normal(mu, 0.1 * mu, N)), 0.5 * mu, 1.5 * mu).astype(int)
        return [r.integers(0, hi, L).astype(np.int32) for L in lens]

    c1 = {}
    vals = synth(1, 8, 100, 20)
    sseqs = [ct.Sequence("s%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, aa, raw_indices=v))])
             for i, v in enumerate(vals)]
    for i, v in enumerate(vals):
        c1["seq%d" % i] = v
    for mode in MODES:
        scores, paths = [], []
        for i in range(8):
            for j in range(i + 1, 8):
                sc, path = pairwise(mode, sseqs[i], sseqs[j], T_IN, [blosum62])
                scores.append(sc)
                paths.append(path)
        c1["scores_" + mode] = np.array(scores, dtype=np.float64)
        c1["paths_" + mode], c1["paths_off_" + mode] = pack_paths(paths)
    save("synthetic_c1.npz", **c1)

    # ---------------------------------------------------------------- DNA (A = 15)
    dna = ct.ALPHABET_DNA
    nuc = load_score_matrix(open_builtin("matrices/nucleotide"), alphabet=dna)
    dn = dict(matrix=nuc.matrix.astype(np.float32))
    dvals = synth(5, 6, 60, 4)
    dseqs = [ct.Sequence("d%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, dna, raw_indices=v))])
             for i, v in enumerate(dvals)]
    for i, v in enumerate(dvals):
        dn["seq%d" % i] = v
    for mode in MODES:
        scores, paths = [], []
        for i in range(6):
            for j in range(i + 1, 6):
                sc, path = pairwise(mode, dseqs[i], dseqs[j], T_IN, [nuc])
                scores.append(sc)
                paths.append(path)
        dn["scores_" + mode] = np.array(scores, dtype=np.float64)
        dn["paths_" + mode], dn["paths_off_" + mode] = pack_paths(paths)
    save("synthetic_dna.npz", **dn)


def make_matrices():
    import shutil
    out = {}
    mdir = os.path.join(REF_ROOT, "praline", "matrices")
    for name in sorted(os.listdir(mdir)):
        alphabet = ct.ALPHABET_DNA if name == "nucleotide" else ct.ALPHABET_AA
        out[name] = load_score_matrix(open_builtin("matrices/" + name), alphabet=alphabet).matrix.astype(np.float32)
    save("matrices.npz", **out)
    for fn in ("BBA0184.cli.aln", "BBA0184.multitrack.aln", "BBA0184.motif.tfa", "BBA0184.ss.tfa", "motif_score_matrix"):
        shutil.copyfile(os.path.join(DATA, fn), os.path.join(HERE, fn))
        print("copied %s" % fn)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "matrices":
        make_matrices()
    else:
        main()
        make_matrices()
