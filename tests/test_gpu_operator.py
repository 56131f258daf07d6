"""GPU tests of the operator layer: the reference's plugin API (type ids, ports, Execution) driving
the HIP path, checked against golden vectors from the real reference."""
import numpy as np
import pytest

from conftest import MODES, load_golden
from oracle import oracle as orc
from praline_amd import component as comp
from praline_amd import container as ct
from praline_amd import core

pytestmark = pytest.mark.gpu
T_IN = [[ct.TRACK_ID_INPUT]]


@pytest.fixture(scope="module")
def env():
    from praline_amd import native
    native.init(0)
    idx = core.TypeIndex()
    idx.autoregister()
    return {"index": idx, "serial": core.Manager(idx), "batch": comp.BatchManager(idx), "blosum": ct.blosum62()}


@pytest.fixture(scope="module")
def seqs(bba):
    return [ct.Sequence("seq%03d" % (i + 1), [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
            for i, v in enumerate(bba["seqs"])]


def run_one(manager, component, env_keys=None, **inputs):
    ex = core.Execution(manager, "root")
    ex.add_task(component).environment(core.Environment({}), core.Environment(dict(env_keys or {}))).inputs(**inputs)
    return core.run(ex)[0]


def test_pairwise_aligner_kats(env, seqs):
    d = load_golden("kat_pairwise.npz")
    for (i, j) in ((0, 1), (0, 4), (2, 3)):
        for mode in MODES:
            out = run_one(env["serial"], comp.PairwiseAligner, mode=mode, sequence_one=seqs[i], sequence_two=seqs[j],
                          track_id_sets_one=T_IN, track_id_sets_two=T_IN, score_matrices=[env["blosum"]])
            assert isinstance(out['score'], float) and out['score'] == float(d["score_%d_%d_%s" % (i, j, mode)])
            path = out['alignment'].path
            if mode.startswith("semiglobal"):
                assert isinstance(path, np.ndarray)
            else:
                assert isinstance(path, list) and isinstance(path[0], tuple)
            assert np.array_equal(np.array(path), d["path_%d_%d_%s" % (i, j, mode)])
            assert out['alignment'].items == [seqs[i], seqs[j]]


def test_batch_manager_is_equivalent(env, seqs):
    d = load_golden("kat_pairwise.npz")
    ex = core.Execution(env["batch"], "root")
    keys = []
    for i in range(5):
        for j in range(i + 1, 5):
            for mode in MODES:
                ex.add_task(comp.PairwiseAligner).environment(core.Environment({}), core.Environment({})).inputs(
                    mode=mode, sequence_one=seqs[i], sequence_two=seqs[j], track_id_sets_one=T_IN,
                    track_id_sets_two=T_IN, score_matrices=[env["blosum"]])
                keys.append((i, j, mode))
    kinds = [m.kind for m in ex.run()]
    assert kinds.count("begin") == len(keys) and kinds.count("complete") == len(keys)
    for (i, j, mode), out in zip(keys, ex.outputs):
        assert out['score'] == float(d["score_%d_%d_%s" % (i, j, mode)])
        assert np.array_equal(np.array(out['alignment'].path), d["path_%d_%d_%s" % (i, j, mode)])


def test_master_slave_aligners_and_profile_builder(env, seqs):
    d = load_golden("preprofile.npz")
    for key, component, master, keys in (("global_m0_", comp.GlobalMasterSlaveAligner, 0, {}),
                                         ("global_m2_", comp.GlobalMasterSlaveAligner, 2, {}),
                                         ("local_m0_", comp.LocalMasterSlaveAligner, 0, {}),
                                         ("local_m4_", comp.LocalMasterSlaveAligner, 4, {}),
                                         ("local_thr_m0_", comp.LocalMasterSlaveAligner, 0, {"score_threshold": 100.0}),
                                         ("local_we3_m0_", comp.LocalMasterSlaveAligner, 0, {"waterman_eggert_iterations": 3})):
        slaves = [s for k, s in enumerate(seqs) if k != master]
        out = run_one(env["serial"], component, keys, master_sequence=seqs[master], slave_sequences=slaves,
                      track_id_sets=T_IN, score_matrices=[env["blosum"]])
        assert np.array_equal(np.asarray(out['alignment'].path), d[key + "msa_path"]), key
        prof = run_one(env["serial"], comp.ProfileBuilder, alignment=out['alignment'], track_id=ct.TRACK_ID_INPUT)
        assert np.array_equal(prof['profile_track'].counts, d[key + "profile_counts"]), key
        assert np.array_equal(prof['profile_track'].profile, d[key + "profile_f32"]), key


def test_device_preprofile_stage(env, seqs, monkeypatch):
    """build_preprofiles: the counting over the master-slave merge runs on the device paths.  Pinned to the
    reference's own ProfileBuilder counts (goldens) and, on a synthetic set with short and empty-ish local
    alignments, to the mirrored component chain.  Both counting kernels: one lane per pair with global atomics (small
    lists) and one workgroup per master with its count block in LDS (PRALINE_COUNT_RUNS=1 forces it; the default from
    4 096 pairs on)."""
    d = load_golden("preprofile.npz")
    blosum = env["blosum"]
    for key, mode, master, kw in (("global_m0_", "global", 0, {}), ("global_m2_", "global", 2, {}),
                                  ("local_m0_", "local", 0, {}), ("local_m4_", "local", 4, {}),
                                  ("local_thr_m0_", "local", 0, {"score_threshold": 100.0}),
                                  ("local_we3_m0_", "local", 0, {"waterman_eggert_iterations": 3})):
        tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode, **kw)
        assert np.array_equal(tracks[master].counts, d[key + "profile_counts"]), key
        assert np.array_equal(tracks[master].profile, d[key + "profile_f32"]), key
    # synthetic: 9 sequences, lengths 3..70, some unrelated (tiny local alignments, starts at slave index 0)
    rng = np.random.default_rng(23)
    lens = [3, 7, 20, 33, 40, 41, 64, 65, 70]
    base = rng.integers(0, 20, 80)
    vals = []
    for n, L in enumerate(lens):
        v = base[:L].copy() if n % 3 else rng.integers(0, 20, L)
        flip = rng.random(L) < 0.15
        v[flip] = rng.integers(0, 20, int(flip.sum()))
        vals.append(v)
    syn = [ct.Sequence("s%d" % n, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
           for n, v in enumerate(vals)]
    for mode, component, kw in (("global", comp.GlobalMasterSlaveAligner, {}),
                                ("local", comp.LocalMasterSlaveAligner, {}),
                                ("local", comp.LocalMasterSlaveAligner, {"score_threshold": 20.0}),
                                ("global", comp.GlobalMasterSlaveAligner, {"score_threshold": 0.0})):
        tracks = comp.build_preprofiles(syn, ct.TRACK_ID_INPUT, blosum, mode=mode, **kw)
        monkeypatch.setenv("PRALINE_COUNT_RUNS", "1")
        tracks_runs = comp.build_preprofiles(syn, ct.TRACK_ID_INPUT, blosum, mode=mode, **kw)
        monkeypatch.delenv("PRALINE_COUNT_RUNS")
        assert all(np.array_equal(a.counts, b.counts) for a, b in zip(tracks, tracks_runs)), (mode, kw)
        for master in range(len(syn)):
            slaves = [s for k, s in enumerate(syn) if k != master]
            out = run_one(env["serial"], component, kw, master_sequence=syn[master], slave_sequences=slaves,
                          track_id_sets=T_IN, score_matrices=[blosum])
            prof = run_one(env["serial"], comp.ProfileBuilder, alignment=out['alignment'], track_id=ct.TRACK_ID_INPUT)
            assert np.array_equal(tracks[master].counts, prof['profile_track'].counts), (mode, kw, master)


def test_guide_tree_on_preprofiles(env, seqs):
    d = load_golden("profile_profile.npz")
    pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)),
                                (ct.TRACK_ID_PREPROFILE, ct.ProfileTrack(d["counts%d" % i], ct.ALPHABET_AA))])
           for i, s in enumerate(seqs)]
    ex = core.Execution(env["serial"], "root")
    ex.add_task(comp.GuideTreeBuilder).environment(core.Environment({}), core.Environment({})).inputs(
        sequences=pre, track_id_sets=[[ct.TRACK_ID_PREPROFILE]], score_matrices=[env["blosum"]])
    tree = core.run(ex)[0]['guide_tree']
    assert [tuple(x) for x in tree.merge_orders] == [tuple(x) for x in d["merge_order"]]
    # profile-profile alignments through the operator API: float scoring within 1e-5 of the reference
    for (i, j) in ((0, 4), (1, 3)):
        for mode in ("global", "local", "semiglobal_both"):
            out = run_one(env["serial"], comp.PairwiseAligner, mode=mode, sequence_one=pre[i], sequence_two=pre[j],
                          track_id_sets_one=[[ct.TRACK_ID_PREPROFILE]], track_id_sets_two=[[ct.TRACK_ID_PREPROFILE]],
                          score_matrices=[env["blosum"]])
            ref = float(d["score_%d_%d_%s" % (i, j, mode)])
            assert abs(out['score'] - ref) <= 1e-5 * max(1.0, abs(ref))
            assert np.array_equal(np.array(out['alignment'].path), d["path_%d_%d_%s" % (i, j, mode)])


def test_guide_tree_distance_modes(env, seqs):
    """GuideTreeBuilder's all-pairs stage (one device submission per alignment mode, no request per pair) against
    one PairwiseAligner execution per pair with the mode the reference picks (tree.py:105-129,
    util/align.py:299-305), for every dist_mode; the merge order follows from the same distances."""
    T = [[ct.TRACK_ID_INPUT]]
    n = len(seqs)
    for dist_mode in ("global", "semiglobal", "semiglobal_auto"):
        ex = core.Execution(env["serial"], "root")
        task = ex.add_task(comp.GuideTreeBuilder)
        task.environment(core.Environment({}), core.Environment({"dist_mode": dist_mode})).inputs(
            sequences=seqs, track_id_sets=T, score_matrices=[env["blosum"]])
        tree = core.run(ex)[0]['guide_tree']
        d = np.zeros((n, n), dtype=np.float32)
        for i in range(n):
            for j in range(i + 1, n):
                if dist_mode == "semiglobal_auto":
                    mode = "semiglobal_one" if len(seqs[i]) > len(seqs[j]) else "semiglobal_two"
                else:
                    mode = {"global": "global", "semiglobal": "semiglobal_both"}[dist_mode]
                out = run_one(env["serial"], comp.PairwiseAligner, mode=mode, sequence_one=seqs[i], sequence_two=seqs[j],
                              track_id_sets_one=T, track_id_sets_two=T, score_matrices=[env["blosum"]])
                d[i, j] = d[j, i] = out['score']
        assert [tuple(x) for x in tree.merge_orders] == comp.merge_order((-d) + d.max(), "average"), dist_mode


def test_multitrack_sets(env, seqs, bba):
    d = load_golden("multitrack.npz")
    motif_alpha = ct.Alphabet("golden.motif", [("*", 0), ("M", 1)])
    ss_alpha = ct.Alphabet("golden.ss", [("C", 0), ("H", 1), ("E", 2)])
    motif_sm = ct.ScoreMatrix(None, [motif_alpha, motif_alpha], matrix=bba["motif_matrix"])
    ss_sm = ct.ScoreMatrix(None, [ss_alpha, ss_alpha], matrix=bba["ss_matrix"])
    ms = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)),
                               ("golden.motif", ct.PlainTrack(None, motif_alpha, raw_indices=bba["motif"][i])),
                               ("golden.ss", ct.PlainTrack(None, ss_alpha, raw_indices=bba["ss"][i]))])
          for i, s in enumerate(seqs)]
    for nsets, tracks, sms in ((2, [[ct.TRACK_ID_INPUT], ["golden.motif"]], [env["blosum"], motif_sm]),
                               (3, [[ct.TRACK_ID_INPUT], ["golden.motif"], ["golden.ss"]], [env["blosum"], motif_sm, ss_sm])):
        for (i, j) in ((0, 1), (0, 4), (2, 3)):
            for mode in ("global", "local", "semiglobal_both"):
                out = run_one(env["serial"], comp.PairwiseAligner, mode=mode, sequence_one=ms[i], sequence_two=ms[j],
                              track_id_sets_one=tracks, track_id_sets_two=tracks, score_matrices=sms)
                k = "s%d_%d_%d_%s_" % (nsets, i, j, mode)
                assert out['score'] == float(d[k + "score"]), k
                assert np.array_equal(np.array(out['alignment'].path), d[k + "path"]), k


def test_raw_pairwise_aligner_and_arbitrary_zero_idxs(env, seqs):
    d = load_golden("fill_small.npz")
    for n in range(int(d["n_cases"])):
        p = "c%03d_" % n
        mode = str(d[p + "mode"])
        m = d[p + "m"]
        a = ct.Sequence("a", [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=np.zeros(m.shape[0], int)))])
        b = ct.Sequence("b", [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=np.zeros(m.shape[1], int)))])
        zi = [tuple(int(v) for v in r) for r in d[p + "zero_idxs"]] if (p + "zero_idxs") in d.files else None
        out = run_one(env["serial"], comp.RawPairwiseAligner, mode=mode, sequence_one=a, sequence_two=b,
                      match_score_model=ct.MatchScoreModel(a, b, m), gap_score_model_one=ct.GapScoreModel(a, d[p + "g1"]),
                      gap_score_model_two=ct.GapScoreModel(b, d[p + "g2"]), zero_idxs=zi)
        assert out['score'] == float(d[p + "score"]) and np.array_equal(np.array(out['alignment'].path), d[p + "path"]), n
    # PairwiseAligner with a scattered (non-rectangular) mask: 60 one-cell rectangles through the batched path
    rng = np.random.default_rng(4)
    zi = [(int(rng.integers(1, 100)), int(rng.integers(1, 100))) for _ in range(60)]
    out = run_one(env["serial"], comp.PairwiseAligner, mode="local", sequence_one=seqs[4], sequence_two=seqs[3],
                  track_id_sets_one=T_IN, track_id_sets_two=T_IN, score_matrices=[env["blosum"]], zero_idxs=zi)
    from conftest import one_hot
    s_ref, p_ref = orc.pairwise_align("local", [one_hot(seqs[4].get_track(ct.TRACK_ID_INPUT).values, 27)],
                                      [one_hot(seqs[3].get_track(ct.TRACK_ID_INPUT).values, 27)], [env["blosum"].matrix],
                                      zero_idxs=zi)
    assert out['score'] == s_ref and np.array_equal(np.array(out['alignment'].path), p_ref)


def test_error_behaviour(env, seqs):
    base = dict(sequence_one=seqs[0], sequence_two=seqs[1], track_id_sets_one=T_IN, track_id_sets_two=T_IN,
                score_matrices=[env["blosum"]])
    with pytest.raises(core.ComponentError):
        run_one(env["serial"], comp.PairwiseAligner, mode="banana", **base)
    with pytest.raises(core.ComponentError):
        run_one(env["serial"], comp.PairwiseAligner, {"gap_series": [-1.0, -2.0, -3.0]}, mode="global", **base)
    with pytest.raises(core.ComponentError):
        run_one(env["serial"], comp.PairwiseAligner, mode="global", **dict(base, track_id_sets_two=[]))
    with pytest.raises(core.ComponentError):
        run_one(env["serial"], comp.PairwiseAligner, mode="global",
                **dict(base, track_id_sets_one=[[ct.TRACK_ID_INPUT, ct.TRACK_ID_INPUT]]))
    with pytest.raises(core.DataError):
        run_one(env["serial"], comp.PairwiseAligner, mode="global", **dict(base, score_matrices=[ct.nucleotide_matrix()]))
    with pytest.raises(core.DataError):
        run_one(env["serial"], comp.PairwiseAligner, mode=3, **base)
    # single gap value = linear gaps (align.py:182-183)
    out = run_one(env["serial"], comp.PairwiseAligner, {"gap_series": [-4.0]}, mode="global", **base)
    from conftest import one_hot
    s_ref, p_ref = orc.pairwise_align("global", [one_hot(seqs[0].get_track(ct.TRACK_ID_INPUT).values, 27)],
                                      [one_hot(seqs[1].get_track(ct.TRACK_ID_INPUT).values, 27)], [env["blosum"].matrix],
                                      gap_series=[-4.0])
    assert out['score'] == s_ref and np.array_equal(np.array(out['alignment'].path), p_ref)


def test_tree_msa_against_reference(env, seqs):
    """GuideTreeBuilder + TreeMultipleSequenceAligner (msa.py:124-237) on the BBA0184 set, preprofile
    tracks (float scoring) and input tracks (integer scoring), three merge modes: every merge step's
    mode, score and path and the final multiple alignment against the real reference's run."""
    d = load_golden("treemsa.npz")
    pp = load_golden("profile_profile.npz")
    pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)),
                                (ct.TRACK_ID_PREPROFILE, ct.ProfileTrack(pp["counts%d" % i], ct.ALPHABET_AA))])
           for i, s in enumerate(seqs)]
    for tag, sset, tracks in (("pre", pre, [[ct.TRACK_ID_PREPROFILE]]), ("in", seqs, T_IN)):
        tree = run_one(env["serial"], comp.GuideTreeBuilder, sequences=sset, track_id_sets=tracks,
                       score_matrices=[env["blosum"]])['guide_tree']
        want_order = pp["merge_order"] if tag == "pre" else d["merge_order_input"]
        assert [tuple(x) for x in tree.merge_orders] == [tuple(x) for x in want_order]
        for merge_mode in ("semiglobal", "global", "semiglobal_auto"):
            key = "%s_%s_" % (tag, merge_mode)
            steps = []
            orig = comp.PairwiseAligner.execute

            def spy(self, *a, _orig=orig, _steps=steps, **kw):
                for msg in _orig(self, *a, **kw):
                    if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.outputs and 'alignment' in msg.outputs:
                        _steps.append((kw.get('mode', a[0] if a else None), msg.outputs['score'],
                                       np.array(msg.outputs['alignment'].path)))
                    yield msg
            comp.PairwiseAligner.execute = spy
            try:
                out = run_one(env["serial"], comp.TreeMultipleSequenceAligner, {"merge_mode": merge_mode},
                              sequences=sset, guide_tree=tree, track_id_sets=tracks, score_matrices=[env["blosum"]])
            finally:
                comp.PairwiseAligner.execute = orig
            assert len(steps) == int(d[key + "n_steps"])
            for c, (mode, score, path) in enumerate(steps):
                assert mode == str(d[key + "step%d_mode" % c]), (key, c)
                ref = float(d[key + "step%d_score" % c])
                # merged clusters are count fractions, so all but single-sequence steps are float scoring
                assert abs(score - ref) <= 1e-5 * abs(ref), (key, c)
                assert np.array_equal(path, d[key + "step%d_path" % c]), (key, c)
            aln = out['alignment']
            assert [s.name for s in aln.items] == [str(x) for x in d[key + "names"]]
            assert np.array_equal(np.asarray(aln.path), d[key + "path"]), key


def test_end_to_end_msa_reproduces_shipped_alignment(env):
    """The reference's only shipped known-answer file, extra/data/BBA0184.aln (committed as data under
    tests/golden/), was produced by: FASTA in -> GlobalMasterSlaveAligner + ProfileBuilder per sequence
    -> GuideTreeBuilder (average linkage, global) on the preprofiles -> TreeMultipleSequenceAligner
    (merge global) -> aligned FASTA out.  The same pipeline on the device path must
    give the same bytes; the preprofile stage is also run in its one-call device form."""
    import os
    from conftest import GOLDEN as GOLDEN_DIR
    from praline_amd import io as pio
    seqs = pio.load_sequence_fasta(os.path.join(GOLDEN_DIR, "BBA0184.tfa"), ct.ALPHABET_AA)
    want = open(os.path.join(GOLDEN_DIR, "BBA0184.aln")).read()
    blosum = env["blosum"]
    for manager, fused in ((env["serial"], False), (env["batch"], True)):
        if fused:
            tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode="global")
        else:
            tracks = []
            for m, master in enumerate(seqs):
                slaves = [s for k, s in enumerate(seqs) if k != m]
                aln = run_one(manager, comp.GlobalMasterSlaveAligner, master_sequence=master, slave_sequences=slaves,
                              track_id_sets=T_IN, score_matrices=[blosum])['alignment']
                tracks.append(run_one(manager, comp.ProfileBuilder, alignment=aln,
                                      track_id=ct.TRACK_ID_INPUT)['profile_track'])
        pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)), (ct.TRACK_ID_PREPROFILE, t)])
               for s, t in zip(seqs, tracks)]
        t_pre = [[ct.TRACK_ID_PREPROFILE]]
        keys = {"linkage_method": "average", "dist_mode": "global", "merge_mode": "global"}
        tree = run_one(manager, comp.GuideTreeBuilder, keys, sequences=pre, track_id_sets=t_pre,
                       score_matrices=[blosum])['guide_tree']
        msa = run_one(manager, comp.TreeMultipleSequenceAligner, keys, sequences=pre, guide_tree=tree,
                      track_id_sets=t_pre, score_matrices=[blosum])['alignment']
        import io
        assert pio.write_alignment_fasta(io.StringIO(), msa, ct.TRACK_ID_INPUT) == want, "fused" if fused else "components"


def test_adhoc_msa_against_reference(env, seqs):
    """AdHocMultipleSequenceAligner (msa.py:250-558) with its score cache: the exact sequence of inner
    alignments (mode, clusters, lengths; scores within 1e-5) under the serial manager, and the final
    alignment under both managers (BatchManager turns every round into one device submission)."""
    d = load_golden("adhoc.npz")
    pp = load_golden("profile_profile.npz")
    pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)),
                                (ct.TRACK_ID_PREPROFILE, ct.ProfileTrack(pp["counts%d" % i], ct.ALPHABET_AA))])
           for i, s in enumerate(seqs)]
    for tag, sset, tracks in (("pre", pre, [[ct.TRACK_ID_PREPROFILE]]), ("in", seqs, T_IN)):
        for merge_mode, dist_mode in (("semiglobal", "global"), ("global", "global"),
                                      ("semiglobal_auto", "semiglobal_auto"), ("global", "semiglobal")):
            key = "%s_%s_%s_" % (tag, merge_mode, dist_mode)
            keys = {"merge_mode": merge_mode, "dist_mode": dist_mode}
            calls = []
            orig = comp.PairwiseAligner.execute

            def spy(self, mode, sequence_one, sequence_two, *a, _orig=orig, _calls=calls, **kw):
                for msg in _orig(self, mode, sequence_one, sequence_two, *a, **kw):
                    if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.outputs and 'alignment' in msg.outputs:
                        _calls.append((mode, sequence_one.name, sequence_two.name, len(sequence_one), len(sequence_two),
                                       msg.outputs['score']))
                    yield msg
            comp.PairwiseAligner.execute = spy
            try:
                out = run_one(env["serial"], comp.AdHocMultipleSequenceAligner, keys, sequences=sset,
                              track_id_sets=tracks, score_matrices=[env["blosum"]])
            finally:
                comp.PairwiseAligner.execute = orig
            assert [c[0] for c in calls] == [str(x) for x in d[key + "call_modes"]], key
            assert [c[1] for c in calls] == [str(x) for x in d[key + "call_one"]], key
            assert [c[2] for c in calls] == [str(x) for x in d[key + "call_two"]], key
            assert np.array_equal(np.array([[c[3], c[4]] for c in calls]), d[key + "call_lens"]), key
            got = np.array([c[5] for c in calls])
            assert np.all(np.abs(got - d[key + "call_scores"]) <= 1e-5 * np.abs(d[key + "call_scores"])), key
            out_b = run_one(env["batch"], comp.AdHocMultipleSequenceAligner, keys, sequences=sset,
                            track_id_sets=tracks, score_matrices=[env["blosum"]])
            for o in (out, out_b):
                assert [s.name for s in o['alignment'].items] == [str(x) for x in d[key + "names"]], key
                assert np.array_equal(np.asarray(o['alignment'].path), d[key + "path"]), key


def _run_component(manager, cls, keys, **inputs):
    """Run a component instance directly (so that its diagnostics stay reachable); returns (instance, outputs)."""
    env_ = core.Environment({}).collapse(cls, core.Environment(dict(keys or {})))
    inst = cls(manager, env_, "root")
    for name, port in cls.inputs.items():
        inputs.setdefault(name, None)
    outputs = None
    for msg in inst.execute(**inputs):
        if msg.kind == core.MESSAGE_KIND_COMPLETE and msg.tag is None:
            outputs = msg.outputs
    return inst, outputs


def test_resident_msa_merges_on_the_device(env, seqs, monkeypatch):
    """under the batching manager the clusters of TreeMultipleSequenceAligner / AdHocMultipleSequenceAligner
    live and grow on the GPU (ResidentClusters): ONE arena, merged clusters appended in place, NO host-side count-track
    merge (ProfileTrack.merge is made to fail).  Every merge step's mode, score (1e-5) and path and the final alignment
    against the real reference's run, preprofile (float) and input (integer) tracks, all merge modes."""
    d = load_golden("treemsa.npz")
    da = load_golden("adhoc.npz")
    pp = load_golden("profile_profile.npz")

    def no_host_merge(self, track, path):
        raise AssertionError("a count track was merged on the host")
    monkeypatch.setattr(ct.ProfileTrack, "merge", no_host_merge)
    arenas = []
    orig_arena = comp.native.Arena

    class CountingArena(orig_arena):
        def __init__(self, *a, **kw):
            arenas.append(1)
            orig_arena.__init__(self, *a, **kw)
    monkeypatch.setattr(comp.native, "Arena", CountingArena)
    pre = [ct.Sequence(s.name, [(ct.TRACK_ID_INPUT, s.get_track(ct.TRACK_ID_INPUT)),
                                (ct.TRACK_ID_PREPROFILE, ct.ProfileTrack(pp["counts%d" % i], ct.ALPHABET_AA))])
           for i, s in enumerate(seqs)]
    for tag, sset, tracks in (("pre", pre, [[ct.TRACK_ID_PREPROFILE]]), ("in", seqs, T_IN)):
        order = pp["merge_order"] if tag == "pre" else d["merge_order_input"]
        tree = ct.SequenceTree(sset, [tuple(int(v) for v in x) for x in order])
        for merge_mode in ("semiglobal", "global", "semiglobal_auto"):
            key = "%s_%s_" % (tag, merge_mode)
            del arenas[:]
            inst, out = _run_component(env["batch"], comp.TreeMultipleSequenceAligner, {"merge_mode": merge_mode},
                                       sequences=sset, guide_tree=tree, track_id_sets=tracks, score_matrices=[env["blosum"]])
            assert len(arenas) == 1, key                      # one arena for the whole progressive alignment
            assert len(inst.steps) == int(d[key + "n_steps"])
            for c, (mode, score, path) in enumerate(inst.steps):
                assert mode == str(d[key + "step%d_mode" % c]), (key, c)
                ref = float(d[key + "step%d_score" % c])
                assert abs(score - ref) <= 1e-5 * abs(ref), (key, c)
                assert np.array_equal(path, d[key + "step%d_path" % c]), (key, c)
            assert [s.name for s in out['alignment'].items] == [str(x) for x in d[key + "names"]]
            assert np.array_equal(np.asarray(out['alignment'].path), d[key + "path"]), key
        for merge_mode, dist_mode in (("semiglobal", "global"), ("global", "global"),
                                      ("semiglobal_auto", "semiglobal_auto"), ("global", "semiglobal")):
            key = "%s_%s_%s_" % (tag, merge_mode, dist_mode)
            del arenas[:]
            out = run_one(env["batch"], comp.AdHocMultipleSequenceAligner, {"merge_mode": merge_mode, "dist_mode": dist_mode},
                          sequences=sset, track_id_sets=tracks, score_matrices=[env["blosum"]])
            assert len(arenas) == 1, key
            assert [s.name for s in out['alignment'].items] == [str(x) for x in da[key + "names"]], key
            assert np.array_equal(np.asarray(out['alignment'].path), da[key + "path"]), key


def test_resident_msa_larger_set_equals_host_path(env):
    """The resident path against the component-by-component host path (serial manager: one PairwiseAligner execution
    and host merges per step) on 24 synthetic sequences with two track sets (amino acids + a 3-letter track): equal
    merge steps (scores to 1e-6: the host path re-derives every cluster's profile from the merged counts like the
    device does) and equal final alignment; the arena grows past its reservation."""
    rng = np.random.default_rng(5)
    n = 24
    base = rng.integers(0, 20, 90)
    seqs2 = []
    for i in range(n):
        L = int(rng.integers(40, 90))
        v = base[:L].copy()
        flip = rng.random(L) < 0.25
        v[flip] = rng.integers(0, 20, int(flip.sum()))
        ss = rng.integers(0, 4, L)
        seqs2.append(ct.Sequence("q%02d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v)),
                                                ("ss", ct.PlainTrack(None, ct.ALPHABET_RNA, raw_indices=ss))]))
    tracks = [[ct.TRACK_ID_INPUT], ["ss"]]
    ss_matrix = ct.ScoreMatrix(None, [ct.ALPHABET_RNA, ct.ALPHABET_RNA],
                               matrix=(np.eye(4, dtype=np.float32) * 3 - 1).astype(np.float32))
    mats = [env["blosum"], ss_matrix]
    tree = run_one(env["batch"], comp.GuideTreeBuilder, sequences=seqs2, track_id_sets=tracks, score_matrices=mats)['guide_tree']
    for merge_mode in ("semiglobal", "global", "semiglobal_auto"):
        host, out_h = _run_component(env["serial"], comp.TreeMultipleSequenceAligner, {"merge_mode": merge_mode},
                                     sequences=seqs2, guide_tree=tree, track_id_sets=tracks, score_matrices=mats)
        dev, out_d = _run_component(env["batch"], comp.TreeMultipleSequenceAligner, {"merge_mode": merge_mode},
                                    sequences=seqs2, guide_tree=tree, track_id_sets=tracks, score_matrices=mats)
        assert len(host.steps) == len(dev.steps) == n - 1
        assert sum(len(lv) for lv in dev.levels) == n - 1 and len(dev.levels) < n - 1   # independent steps were batched
        for (m1, s1, p1), (m2, s2, p2) in zip(host.steps, dev.steps):
            assert m1 == m2 and abs(s1 - s2) <= 1e-6 * max(1.0, abs(s1)) and np.array_equal(p1, p2)
        assert np.array_equal(np.asarray(out_h['alignment'].path), np.asarray(out_d['alignment'].path))
    out_h = run_one(env["serial"], comp.AdHocMultipleSequenceAligner, sequences=seqs2[:10], track_id_sets=tracks, score_matrices=mats)
    out_d = run_one(env["batch"], comp.AdHocMultipleSequenceAligner, sequences=seqs2[:10], track_id_sets=tracks, score_matrices=mats)
    assert np.array_equal(np.asarray(out_h['alignment'].path), np.asarray(out_d['alignment'].path))


def test_preprofile_counts_through_rccl_in_place(env, seqs):
    """The multi-GPU branch of build_preprofiles on hardware: the counts are accumulated straight into a torch-owned
    device tensor (praline_arena_counts_bind) and all-reduced in place over RCCL.  One process owns one GPU here, so the
    group has a single rank and the all-reduce is the identity: the tracks must equal the single-rank call's (the
    sharding itself is covered by the gloo world-2/3 tests)."""
    import os
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from praline_amd import allpairs
        want = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, env["blosum"], mode="local", waterman_eggert_iterations=2)
        profiles = [comp._track_profile(s.get_track(ct.TRACK_ID_INPUT)) for s in seqs]
        S = np.ascontiguousarray(env["blosum"].matrix, dtype=np.float32)
        n = len(seqs)
        pairs = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32)
        counts = comp._preprofile_counts_exchange(profiles, S, pairs, "local", -11.0, -1.0, None, 2, world=2, group=None)
        off = 0
        for i, s in enumerate(seqs):
            L = len(s)
            c = counts[off:off + L].astype(int).copy()
            c[np.arange(L), np.asarray(s.get_track(ct.TRACK_ID_INPUT).values)] += 1
            assert np.array_equal(c, np.asarray(want[i].counts)), i
            off += L
        # and the score exchange helper on the same one-rank group
        sc = allpairs.all_gather_scores(np.arange(5, dtype=np.float32), [np.arange(5)], 0, 1, None)
        assert np.array_equal(sc, np.arange(5, dtype=np.float32))
    finally:
        if created:
            dist.destroy_process_group()


def _multitrack_inputs():
    """BBA0184 with its annotation tracks, read from the reference's own input files (tests/golden: byte copies of
    extra/data) through this package's loaders."""
    import os
    from conftest import GOLDEN as GOLDEN_DIR
    from praline_amd import io as pio
    g = lambda fn: os.path.join(GOLDEN_DIR, fn)
    seqs = pio.load_sequence_fasta(g("BBA0184.tfa"), ct.ALPHABET_AA)
    blosum = pio.load_score_matrix(pio.open_builtin("matrices/blosum62"), alphabet=ct.ALPHABET_AA)
    return g, seqs, blosum


def _tree_msa(manager, seqs, track_id_sets, score_matrices):
    keys = {"gap_series": [-11.0, -1.0], "linkage_method": "average", "dist_mode": "global", "merge_mode": "global"}
    tree = run_one(manager, comp.GuideTreeBuilder, keys, sequences=seqs, track_id_sets=track_id_sets,
                   score_matrices=score_matrices)['guide_tree']
    return run_one(manager, comp.TreeMultipleSequenceAligner, keys, sequences=seqs, guide_tree=tree,
                   track_id_sets=track_id_sets, score_matrices=score_matrices)['alignment']


def test_end_to_end_multitrack_cli_alignment(env):
    """extra/data/BBA0184.cli.aln, the reference's second shipped known-answer file, byte for byte on the device path:
    `multitrack_cli.py BBA0184.tfa out -a BBA0184.motif.tfa:motif_score_matrix` (extra/multitrack_cli.py:40-193) = the
    annotation score matrix loaded WITHOUT an alphabet (one is made from its header), the annotation FASTA read with
    that alphabet, dummy master-slave alignments -> ProfileBuilder, then guide tree + TreeMSA over the two track sets
    [preprofile] and [motif] (merge / dist global, average linkage), FASTA out."""
    import io
    from praline_amd import io as pio
    g, seqs, blosum = _multitrack_inputs()
    want = open(g("BBA0184.cli.aln")).read()
    motif_sm = pio.load_score_matrix(g("motif_score_matrix"))
    trid = "praline.example.CustomTrackFile_motif_score_matrix"
    for s, a in zip(seqs, pio.load_sequence_fasta(g("BBA0184.motif.tfa"), motif_sm.alphabets[0])):
        s.add_track(trid, a.get_track(ct.TRACK_ID_INPUT))
    for manager in (env["serial"], env["batch"]):
        for m, master in enumerate(seqs):
            slaves = [s for k, s in enumerate(seqs) if k != m]
            aln = run_one(manager, comp.DummyMasterSlaveAligner, master_sequence=master, slave_sequences=slaves,
                          track_id_sets=[[ct.TRACK_ID_INPUT], [trid]], score_matrices=[blosum, motif_sm])['alignment']
            track = run_one(manager, comp.ProfileBuilder, alignment=aln, track_id=ct.TRACK_ID_INPUT)['profile_track']
            if ct.TRACK_ID_PREPROFILE in dict(master.tracks):
                master.replace_track(ct.TRACK_ID_PREPROFILE, track)
            else:
                master.add_track(ct.TRACK_ID_PREPROFILE, track)
        msa = _tree_msa(manager, seqs, [[ct.TRACK_ID_PREPROFILE], [trid]], [blosum, motif_sm])
        assert pio.write_alignment_fasta(io.StringIO(), msa, ct.TRACK_ID_INPUT) == want, type(manager).__name__


def test_end_to_end_multitrack_notebook_alignment(env):
    """extra/data/BBA0184.multitrack.aln (extra/MSAMultiTrack.ipynb), byte for byte: three track sets - residues (global
    master-slave preprofiles), motif matches (15 / 0) and three-state secondary structure (3 / 0) - through
    GlobalMasterSlaveAligner -> ProfileBuilder -> GuideTreeBuilder -> TreeMultipleSequenceAligner: num_sets = 3 in the
    match-score build of every alignment (cext.c:389-420 sums the sets)."""
    import io
    from praline_amd import io as pio
    g, seqs, blosum = _multitrack_inputs()
    want = open(g("BBA0184.multitrack.aln")).read()
    a_motif = ct.Alphabet("praline.example.SimpleMotifMatch", [("*", 0), ("M", 1)])
    a_ss = ct.Alphabet("praline.example.ThreeStateSecondaryStructure", [("C", 0), ("H", 1), ("E", 2)])
    sm_motif = ct.ScoreMatrix({("M", "M"): 15, ("M", "*"): 0, ("*", "M"): 0, ("*", "*"): 0}, [a_motif, a_motif])
    sm_ss = ct.ScoreMatrix({(a, b): (3 if a == b else 0) for a in "CHE" for b in "CHE"}, [a_ss, a_ss])
    t_motif, t_ss = "praline.example.MotifTrack", "praline.example.SecondaryStructureTrack"
    for s, m, q in zip(seqs, pio.load_sequence_fasta(g("BBA0184.motif.tfa"), a_motif), pio.load_sequence_fasta(g("BBA0184.ss.tfa"), a_ss)):
        s.add_track(t_motif, m.get_track(ct.TRACK_ID_INPUT))
        s.add_track(t_ss, q.get_track(ct.TRACK_ID_INPUT))
    sms = [blosum, sm_motif, sm_ss]
    for manager in (env["serial"], env["batch"]):
        tracks = []
        for m, master in enumerate(seqs):
            slaves = [s for k, s in enumerate(seqs) if k != m]
            aln = run_one(manager, comp.GlobalMasterSlaveAligner, {"gap_series": [-11.0, -1.0]}, master_sequence=master,
                          slave_sequences=slaves, track_id_sets=[[ct.TRACK_ID_INPUT], [t_motif], [t_ss]],
                          score_matrices=sms)['alignment']
            tracks.append(run_one(manager, comp.ProfileBuilder, alignment=aln, track_id=ct.TRACK_ID_INPUT)['profile_track'])
        for s, t in zip(seqs, tracks):
            if ct.TRACK_ID_PREPROFILE in dict(s.tracks):
                s.replace_track(ct.TRACK_ID_PREPROFILE, t)
            else:
                s.add_track(ct.TRACK_ID_PREPROFILE, t)
        msa = _tree_msa(manager, seqs, [[ct.TRACK_ID_PREPROFILE], [t_motif], [t_ss]], sms)
        assert pio.write_alignment_fasta(io.StringIO(), msa, ct.TRACK_ID_INPUT) == want, type(manager).__name__


def test_preprofile_stage_behind_execute_many(env, seqs, monkeypatch):
    """The reference's workflow hands Manager.execute_many ONE list with a Global / LocalMasterSlaveAligner task per master
    and then one with a ProfileBuilder task per master (praline/component/workflow.py:139-161, 211-224).  Under
    BatchManager the first list is one arena, one path plan over every (master, slave) pair and one run per
    Waterman-Eggert iteration (the masks are built on the device), the second is counted in bulk - with the outputs of the
    reference's own run (tests/golden/preprofile.npz) and of the serial manager."""
    from praline_amd import native
    d = load_golden("preprofile.npz")
    runs, plans = [], []
    orig_run, orig_init = native.Plan.run, native.Plan.__init__
    monkeypatch.setattr(native.Plan, "run", lambda self, *a, **k: (runs.append(1), orig_run(self, *a, **k))[1])
    monkeypatch.setattr(native.Plan, "__init__", lambda self, *a, **k: (plans.append(1), orig_init(self, *a, **k))[1])
    for component, keys, golden, iterations in ((comp.GlobalMasterSlaveAligner, {}, {0: "global_m0_", 2: "global_m2_"}, 1),
                                                (comp.LocalMasterSlaveAligner, {}, {0: "local_m0_", 4: "local_m4_"}, 2),
                                                (comp.LocalMasterSlaveAligner, {"score_threshold": 100.0}, {0: "local_thr_m0_"}, 2),
                                                (comp.LocalMasterSlaveAligner, {"waterman_eggert_iterations": 3}, {0: "local_we3_m0_"}, 3)):
        outs = {}
        for name in ("batch", "serial"):
            ex = core.Execution(env[name], "root")
            for m, master in enumerate(seqs):
                ex.add_task(component).environment(core.Environment({}), core.Environment(dict(keys))).inputs(
                    master_sequence=master, slave_sequences=[s for k, s in enumerate(seqs) if k != m],
                    track_id_sets=T_IN, score_matrices=[env["blosum"]])
            del runs[:], plans[:]
            alignments = [o['alignment'] for o in core.run(ex)]
            if name == "batch":
                assert len(plans) == 1 and len(runs) == iterations, (component.__name__, keys, len(plans), len(runs))
            else:
                assert len(runs) >= len(seqs)       # (the serial manager: at least one submission per master)
            ex = core.Execution(env[name], "root")
            for aln in alignments:
                ex.add_task(comp.ProfileBuilder).environment(core.Environment({}), core.Environment({})).inputs(
                    alignment=aln, track_id=ct.TRACK_ID_INPUT)
            del runs[:]
            tracks = [o['profile_track'] for o in core.run(ex)]
            assert not runs                         # counting needs no device work
            outs[name] = (alignments, tracks)
        for m in range(len(seqs)):
            assert np.array_equal(np.asarray(outs["batch"][0][m].path), np.asarray(outs["serial"][0][m].path)), (keys, m)
            assert np.array_equal(outs["batch"][1][m].counts, outs["serial"][1][m].counts), (keys, m)
        for m, key in golden.items():
            assert np.array_equal(np.asarray(outs["batch"][0][m].path), d[key + "msa_path"]), key
            assert np.array_equal(outs["batch"][1][m].counts, d[key + "profile_counts"]), key
            assert np.array_equal(outs["batch"][1][m].profile, d[key + "profile_f32"]), key
    # a mixed list (two component kinds) and a user's own aligner stay on the reference's per-task path
    ex = core.Execution(env["batch"], "root")
    ex.add_task(comp.GlobalMasterSlaveAligner).environment(core.Environment({}), core.Environment({})).inputs(
        master_sequence=seqs[0], slave_sequences=seqs[1:], track_id_sets=T_IN, score_matrices=[env["blosum"]])
    ex.add_task(comp.ProfileBuilder).environment(core.Environment({}), core.Environment({})).inputs(
        alignment=outs["serial"][0][0], track_id=ct.TRACK_ID_INPUT)
    mixed = core.run(ex)
    assert np.array_equal(np.asarray(mixed[0]['alignment'].path), d["global_m0_msa_path"])
    assert np.array_equal(mixed[1]['profile_track'].counts, outs["serial"][1][0].counts)


def test_score_shard_stays_on_the_device_through_the_exchange(env, seqs):
    """Under RCCL the all-pairs stage keeps its score shard on the device: PairwiseBatch.scores_for_pairs(on_device=True)
    lets the kernels write into a torch tensor, allpairs.all_gather_scores gathers and reorders it there and copies the
    complete list once (the benchmarked path, bench.py).  One rank here (a one-rank nccl group exercises the RCCL call):
    the result equals the host path's bit for bit, for one mode and for a mixed-mode list."""
    import os
    import torch
    import torch.distributed as dist
    from praline_amd import allpairs
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert allpairs.group_is_nccl(None)
        n = len(seqs)
        ii, jj = np.triu_indices(n, k=1)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        shards = allpairs.shard_columns(lens, np.stack([ii, jj], axis=1), 1)
        for modes in (np.array(["global"] * len(ii)), np.array(["global", "semiglobal_both", "local"])[np.arange(len(ii)) % 3]):
            batch = comp.PairwiseBatch(T_IN, T_IN, [env["blosum"]], [-11.0, -1.0])
            mine = shards[0]
            host = batch.scores_for_pairs(seqs, ii[mine], jj[mine], modes[mine])
            dev = batch.scores_for_pairs(seqs, ii[mine], jj[mine], modes[mine], on_device=True)
            assert isinstance(dev, torch.Tensor) and dev.is_cuda
            assert np.array_equal(dev.cpu().numpy(), host)
            full = allpairs.all_gather_scores(dev, shards, 0, 1, None)
            want = np.zeros(len(ii), dtype=np.float32)
            want[mine] = host
            assert np.array_equal(full, want)
    finally:
        if created:
            dist.destroy_process_group()


def test_all_pairs_scores_with_the_schedule_prepared_beside_the_arena(env):
    """PairwiseBatch.scores_for_pairs on a one-mode list of 1024 pairs and more (a guide tree's all-pairs stage)
    schedules the plan on a second host thread while the arena is created (native.prepare_schedule_async): the scores
    equal those of plans scheduled the ordinary way - for profile tracks (the prepared pipeline schedule is used) and for
    plain sequences (their kernels take another kind of schedule: the prepared one is dropped)."""
    from praline_amd import native
    rng = np.random.default_rng(77)
    n = 130
    plain, profile = [], []
    for i in range(n):
        L = int(rng.integers(60, 180))
        idx = rng.integers(0, 20, L)
        plain.append(ct.Sequence("p%03d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=idx))]))
        counts = np.zeros((L, 27), dtype=int)
        counts[np.arange(L), idx] += 3
        counts[np.arange(L), rng.integers(0, 20, L)] += rng.integers(0, 3, L)
        profile.append(ct.Sequence("q%03d" % i, [(ct.TRACK_ID_INPUT, ct.ProfileTrack(counts, ct.ALPHABET_AA))]))
    ii, jj = np.triu_indices(n, k=1)
    assert len(ii) >= 1024
    modes = np.array(["global"] * len(ii))
    calls = []
    real = native.prepare_schedule_async

    def counted(lens, pairs):
        calls.append(len(pairs))
        return real(lens, pairs)

    native.prepare_schedule_async = counted
    try:
        for seqs_ in (profile, plain):
            batch = comp.PairwiseBatch(T_IN, T_IN, [env["blosum"]], [-11.0, -1.0])
            got = batch.scores_for_pairs(seqs_, ii, jj, modes)
            # the same list in two modes takes the ordinary road (one plan per mode)
            half = np.array(["global", "local"])[(np.arange(len(ii)) % 2)]
            batch2 = comp.PairwiseBatch(T_IN, T_IN, [env["blosum"]], [-11.0, -1.0])
            mixed = batch2.scores_for_pairs(seqs_, ii, jj, half)
            sel = half == "global"
            assert np.array_equal(got[sel], mixed[sel])
            for k in rng.choice(len(ii), 6, replace=False):
                out = run_one(env["serial"], comp.PairwiseAligner, mode="global", sequence_one=seqs_[ii[k]], sequence_two=seqs_[jj[k]],
                              track_id_sets_one=T_IN, track_id_sets_two=T_IN, score_matrices=[env["blosum"]])
                assert np.float32(out['score']) == got[k]
    finally:
        native.prepare_schedule_async = real
    assert calls == [len(ii), len(ii)]


def test_adhoc_join_order_on_a_near_tie(env):
    """A case scripts/stress_msa.py found (tests/golden/adhoc_tie_case.json: nine short sequences, two track sets): two cluster
    pairs whose scores differ by ~1e-7.  The batching manager scores cluster pairs with scores-only plans (f16 hi/lo split),
    the serial one with single alignments (fp32 chain) - in the default match-score mode the two may join the clusters in a
    different order; in the reference-order mode (bit-identical scores on every path) they must give the same alignment."""
    import json, os
    from praline_amd import native
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adhoc_tie_case.json")))
    seqs = []
    for i, tracks in enumerate(d["tracks"]):
        seqs.append(ct.Sequence("q%02d" % i, [(tid, ct.PlainTrack(None, ct.ALPHABET_AA if tid == ct.TRACK_ID_INPUT else ct.ALPHABET_RNA,
                                                               raw_indices=np.array(vals))) for tid, vals in tracks]))
    ss = ct.ScoreMatrix(None, [ct.ALPHABET_RNA, ct.ALPHABET_RNA], matrix=(np.eye(4, dtype=np.float32) * 3 - 1).astype(np.float32))
    native.set_match_mode("ref")
    try:
        outs = []
        for name in ("batch", "serial"):
            outs.append(run_one(env[name], comp.AdHocMultipleSequenceAligner, {"merge_mode": d["merge_mode"], "dist_mode": d["dist_mode"]},
                                sequences=seqs, track_id_sets=[[ct.TRACK_ID_INPUT], ["ss"]], score_matrices=[env["blosum"], ss])['alignment'])
    finally:
        native.set_match_mode(None)
    assert [x.name for x in outs[0].items] == [x.name for x in outs[1].items]
    assert np.array_equal(np.asarray(outs[0].path), np.asarray(outs[1].path))
