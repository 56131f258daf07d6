"""CPU tests of the host-side mirror of the reference interface (no GPU, no native compute):
core runtime semantics, containers, path utilities and the master-slave merge logic, checked against
the golden vectors generated from the real reference."""
import os

import numpy as np
import pytest

from conftest import load_golden
from praline_amd import component as comp
from praline_amd import container as ct
from praline_amd import core, util


def make_seqs(bba):
    return [ct.Sequence("seq%03d" % (i + 1), [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
            for i, v in enumerate(bba["seqs"])]


def test_alphabets_and_matrices(bba):
    assert ct.ALPHABET_AA.size == 27 and ct.ALPHABET_DNA.size == 15 and ct.ALPHABET_RNA.size == 4
    assert [ct.ALPHABET_AA.symbol_to_index(c) for c in "ARNDCEQ*"] == [0, 1, 2, 3, 4, 5, 6, 26]
    assert np.array_equal(ct.blosum62().matrix, bba["S"])
    assert np.array_equal(ct.nucleotide_matrix().matrix, load_golden("synthetic_dna.npz")["matrix"])
    with pytest.raises(core.AlphabetError):
        ct.ALPHABET_AA.symbol_to_index("?")
    t = ct.PlainTrack("ARN*", ct.ALPHABET_AA)
    assert t.values.dtype == np.int32 and list(t.values) == [0, 1, 2, 26]


def test_sequence_track_rules():
    a = ct.PlainTrack("ARND", ct.ALPHABET_AA)
    s = ct.Sequence("s", [(ct.TRACK_ID_INPUT, a)])
    assert len(s) == 4
    with pytest.raises(core.DataError):
        s.add_track(ct.TRACK_ID_INPUT, a)
    with pytest.raises(core.DataError):
        s.add_track("other", ct.PlainTrack("AR", ct.ALPHABET_AA))
    with pytest.raises(core.DataError):
        s.get_track("missing")
    s.replace_track(ct.TRACK_ID_INPUT, ct.PlainTrack("AR", ct.ALPHABET_AA))
    assert len(s) == 2


def test_profile_track_normalisation_and_merge():
    d = load_golden("profile_profile.npz")
    for i in range(5):
        tr = ct.ProfileTrack(d["counts%d" % i], ct.ALPHABET_AA)
        assert np.array_equal(tr.profile, d["profile%d" % i])  # sequence.py:200-202
    a = ct.ProfileTrack([[1, 0, 2], [0, 3, 0]], ct.Alphabet("x", [("a", 0), ("b", 1), ("c", 2)]))
    b = ct.ProfileTrack([[0, 1, 0]], a.alphabet)
    merged = a.merge(b, np.array([[0, 0], [1, 0], [2, 1]]))
    assert np.array_equal(merged.counts, [[1, 0, 2], [0, 4, 0]])


def test_environment_inheritance():
    class C(core.Component):
        options = {'x': int, 'sub': core.Environment.tid}
        defaults = {'x': 1, 'y': 5, 'sub': core.Environment({'a': 1})}

    parent = core.Environment({'x': 2, 'sub': core.Environment({'b': 2})})
    env = parent.collapse(C, core.Environment({'z': 3, 'sub': core.Environment({'c': 3})}))
    assert env['x'] == 2 and env['y'] == 5 and env['z'] == 3
    assert env['sub'].keys == {'a': 1, 'b': 2, 'c': 3}


def test_manager_checks_ports_and_options():
    class Echo(core.Component):
        tid = "test.Echo"
        inputs = {'value': core.Port(int), 'opt': core.Port(str, optional=True)}
        outputs = {'value': core.Port(int)}
        options = {'k': int}
        defaults = {'k': 1}

        def execute(self, value, opt):
            yield core.ProgressMessage(0.5)
            yield core.CompleteMessage({'value': value + self.environment['k']})

    idx = core.TypeIndex()
    idx.register(Echo)
    man = core.Manager(idx)
    ex = core.Execution(man, "root")
    ex.add_task(Echo).environment(core.Environment({}), core.Environment({'k': 4})).inputs(value=3)
    kinds = [m.kind for m in ex.run()]
    assert kinds == ["begin", "progress", "complete"] and ex.outputs[0]['value'] == 7
    ex = core.Execution(man, "root")
    ex.add_task(Echo).inputs(value="no")
    with pytest.raises(core.DataError):
        list(ex.run())
    ex = core.Execution(man, "root")
    ex.add_task(Echo).inputs()
    with pytest.raises(core.DataError):
        list(ex.run())
    with pytest.raises(core.ComponentError):
        idx.resolve("nope")
    with pytest.raises(core.ComponentError):
        core.Execution(man).outputs
    with pytest.raises(core.SignatureError):
        core.check_signature([int, str])
    man.close()
    with pytest.raises(core.PralineError):
        list(man.execute_many([], None))


def test_component_declarations_match_reference():
    """Type ids, ports, options and defaults of the hot-path components (align.py:75-86,289-300;
    profile.py:32-39)."""
    pa = comp.PairwiseAligner
    assert pa.tid == "praline.component.PairwiseAligner"
    assert set(pa.inputs) == {'mode', 'sequence_one', 'sequence_two', 'track_id_sets_one',
                              'track_id_sets_two', 'zero_idxs', 'score_matrices'}
    assert pa.inputs['zero_idxs'].optional and not pa.inputs['mode'].optional
    assert set(pa.outputs) == {'alignment', 'score'}
    assert pa.defaults == {'gap_series': [-11.0, -1.0], 'debug': 0}
    raw = comp.RawPairwiseAligner
    assert raw.tid == "praline.component.RawPairwiseAligner"
    assert set(raw.inputs) == {'mode', 'sequence_one', 'sequence_two', 'match_score_model',
                               'gap_score_model_one', 'gap_score_model_two', 'zero_idxs'}
    assert raw.defaults == {'debug': 0, 'accelerate': True}
    pb = comp.ProfileBuilder
    assert pb.tid == "praline.component.ProfileBuilder" and set(pb.inputs) == {'alignment', 'track_id'}
    tm = comp.TreeMultipleSequenceAligner    # msa.py:56-69
    assert tm.tid == "praline.component.TreeMultipleSequenceAligner"
    assert set(tm.inputs) == {'sequences', 'guide_tree', 'track_id_sets', 'score_matrices'}
    assert tm.defaults['merge_mode'] == 'semiglobal' and tm.defaults['aligner'] == pa.tid
    ah = comp.AdHocMultipleSequenceAligner   # msa.py:289-301
    assert ah.tid == "praline.component.AdHocMultipleSequenceAligner"
    assert set(ah.inputs) == {'sequences', 'track_id_sets', 'score_matrices'}
    assert ah.defaults['merge_mode'] == 'semiglobal' and ah.defaults['dist_mode'] == 'global'
    idx = core.TypeIndex()
    idx.autoregister()
    for cls in comp.COMPONENTS:
        assert idx.resolve(cls.tid) is cls


def test_path_utilities_and_master_slave_merge(bba):
    """compress_path / extend_path_local / Alignment.merge / get_frequencies (util/align.py:187-266,
    container/align.py:30-61): feed the reference's own inner alignment results (golden) through the
    host logic and compare the merged master-slave alignment and the ProfileBuilder counts."""
    d = load_golden("preprofile.npz")
    seqs = make_seqs(bba)
    for key, master, local, iters, thr in (("global_m0_", 0, False, 1, None), ("global_m2_", 2, False, 1, None),
                                           ("global_m4_", 4, False, 1, None), ("local_m0_", 0, True, 2, None),
                                           ("local_m4_", 4, True, 2, None), ("local_thr_m0_", 0, True, 2, 100.0),
                                           ("local_we3_m0_", 0, True, 3, None)):
        slaves = [k for k in range(5) if k != master]
        results, c = [], 0
        for _ in slaves:
            res = []
            for _ in range(iters):
                res.append((float(d[key + "call%d_score" % c]), d[key + "call%d_path" % c]))
                c += 1
            results.append(res)
        assert c == int(d[key + "n_calls"])
        aln = comp.merge_master_slave(seqs[master], [seqs[k] for k in slaves], results, thr, local)
        assert np.array_equal(np.asarray(aln.path), d[key + "msa_path"]), key
        assert [s.name for s in aln.items] == [str(x) for x in d[key + "msa_names"]]
        freqs = util.get_frequencies(aln, ct.TRACK_ID_INPUT)
        assert np.array_equal(freqs, d[key + "profile_counts"]), key
        out = list(comp.ProfileBuilder(None, core.Environment({'debug': 0}), "t").execute(aln, ct.TRACK_ID_INPUT))[-1].outputs
        assert np.array_equal(out['profile_track'].profile, d[key + "profile_f32"]), key


def test_extend_path_semiglobal_matches_oracle():
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    for _ in range(50):
        n, m = (int(v) for v in rng.integers(2, 30, 2))
        y0, x0 = (int(rng.integers(0, n)), 0) if rng.random() < 0.5 else (0, int(rng.integers(0, m)))
        path = [(y0, x0)]
        while path[-1][0] < n - 1 and path[-1][1] < m - 1 and rng.random() < 0.9:
            y, x = path[-1]
            step = rng.integers(0, 3)
            path.append((y + (step != 2), x + (step != 1)))
        path = np.array(path)
        assert np.array_equal(util.extend_path_semiglobal(path, (n, m)), orc.extend_path_semiglobal(path, (n, m)))


def test_zero_idxs_rectangles():
    zi = [(y, x) for y in range(3, 7) for x in range(10, 15)] + [(y, x) for y in range(1, 3) for x in range(2, 4)]
    assert util.zero_idxs_to_rectangles(zi) == [(3, 6, 10, 14), (1, 2, 2, 3)]
    assert util.zero_idxs_to_rectangles([(1, 1), (3, 3), (5, 5)]) == [(1, 1, 1, 1), (3, 3, 3, 3), (5, 5, 5, 5)]
    assert util.zero_idxs_to_rectangles([(1, 1), (3, 3), (5, 5), (7, 7), (9, 9), (11, 11)], max_rects=4) is None
    assert util.zero_idxs_to_rectangles([]) == []
    # a parsed rectangle list always reproduces the same mask
    rng = np.random.default_rng(1)
    for _ in range(20):
        rects = []
        for _ in range(int(rng.integers(1, 4))):
            y0, x0 = (int(v) for v in rng.integers(0, 20, 2))
            rects.append((y0, y0 + int(rng.integers(0, 6)), x0, x0 + int(rng.integers(0, 6))))
        zi = [(y, x) for (a, b, c, e) in rects for y in range(a, b + 1) for x in range(c, e + 1)]
        parsed = util.zero_idxs_to_rectangles(zi)
        z1 = np.zeros((40, 40), bool)
        z2 = np.zeros((40, 40), bool)
        for idx in zi:
            z1[idx] = True
        for (a, b, c, e) in parsed:
            z2[a:b + 1, c:e + 1] = True
        assert np.array_equal(z1, z2)


def test_merge_order_matches_reference(bba):
    d = load_golden("profile_profile.npz")
    assert [tuple(x) for x in d["merge_order"]] == comp.merge_order(d["dist"], "average")
    dist = np.array([[0, 2, 6, 10], [2, 0, 5, 9], [6, 5, 0, 4], [10, 9, 4, 0]], float)
    assert comp.merge_order(dist, "single") == [(0, 1), (2, 3), (0, 2)]
    assert comp.merge_order(dist, "complete") == [(0, 1), (2, 3), (0, 2)]


def test_all_pairs_sharding_helpers():
    from praline_amd import allpairs
    pairs = allpairs.enumerate_pairs(5)
    assert [tuple(p) for p in pairs][:5] == [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2)] and len(pairs) == 10
    cells = np.array([5, 1, 1, 1, 1, 1, 5, 5, 1, 1])
    for world in (1, 2, 3, 8, 16):
        b = allpairs.shard_bounds(cells, world)
        assert b[0] == 0 and b[-1] == len(cells) and all(b[i] <= b[i + 1] for i in range(world))
    # exchange maps: padded slices concatenated rank by rank -> row-major pair order
    lens9 = np.array([30, 80, 45, 61, 12, 99, 7, 50, 33])
    pairs9 = allpairs.enumerate_pairs(9)
    truth = np.arange(len(pairs9), dtype=np.float32) * 1.5
    for world in (1, 2, 3, 8):
        shards9 = allpairs.shard_columns(lens9, pairs9, world)
        src, dst, shard_len = allpairs.gather_maps(shards9)
        gathered = np.full(shard_len * world, np.nan, dtype=np.float32)
        for r, ix in enumerate(shards9):
            gathered[r * shard_len:r * shard_len + len(ix)] = truth[ix]
        out = np.zeros(len(pairs9), dtype=np.float32)
        out[dst] = gathered[src]
        assert np.array_equal(out, truth), world
    # column shards: a partition of the pair list, whole columns per rank, balanced by cells
    rng = np.random.default_rng(4)
    lens = rng.integers(50, 500, 40)
    pairs40 = allpairs.enumerate_pairs(40)
    cells40 = lens[pairs40[:, 0]].astype(np.int64) * lens[pairs40[:, 1]]
    for world in (1, 2, 3, 8):
        shards = allpairs.shard_columns(lens, pairs40, world)
        assert len(shards) == world
        allidx = np.concatenate(shards)
        assert np.array_equal(np.sort(allidx), np.arange(len(pairs40)))
        owner_of_col = {}
        for r, ix in enumerate(shards):
            assert np.all(np.diff(ix) > 0)
            for j in np.unique(pairs40[ix, 1]):
                assert owner_of_col.setdefault(int(j), r) == r
        loads = np.array([cells40[ix].sum() for ix in shards], dtype=np.float64)
        assert loads.max() <= 1.15 * loads.mean()
    d, dist = allpairs.scores_to_distance(3, [(0, 1), (0, 2), (1, 2)], np.array([5.0, -2.0, 1.0], np.float32))
    assert d[0, 1] == d[1, 0] == 5.0 and d[1, 1] == 0.0 and np.array_equal(dist, (-d) + 5.0)


def test_fasta_io_round_trip(tmp_path):
    """praline_amd.io: reader (names, upper-casing, wrapped lines) and aligned-FASTA writer (72 columns,
    '-' where a sequence does not advance) on the shipped data file."""
    import os
    from conftest import GOLDEN as GOLDEN_DIR
    from praline_amd import io as pio, container as ct
    seqs = pio.load_sequence_fasta(os.path.join(GOLDEN_DIR, "BBA0184.tfa"), ct.ALPHABET_AA)
    assert [s.name for s in seqs] == ["seq%03d" % i for i in range(1, 6)]
    assert [len(s) for s in seqs] == [326, 335, 304, 305, 118]
    a, b = seqs[4], seqs[2]
    path = np.array([(0, 0), (1, 1), (2, 1), (2, 2), (3, 3)])
    aln = ct.Alignment([a, b], path)
    rows = pio.alignment_rows(aln)
    sym = lambda s, k: ct.ALPHABET_AA.index_to_symbol(int(s.get_track(ct.TRACK_ID_INPUT).values[k]))
    assert rows[0] == sym(a, 0) + sym(a, 1) + "-" + sym(a, 2)
    assert rows[1] == sym(b, 0) + "-" + sym(b, 1) + sym(b, 2)
    out = tmp_path / "x.aln"
    text = pio.write_alignment_fasta(str(out), aln)
    assert out.read_text() == text == ">seq005\n%s\n>seq003\n%s\n" % (rows[0], rows[1])
    # the shipped alignment parses back into the shipped sequences once the gaps are removed
    gapped = ct.Alphabet("gapped", [(sy, ct.ALPHABET_AA.symbol_to_index(sy)) for sy in ct.ALPHABET_AA.symbols] + [("-", 99)])
    aligned = pio.load_sequence_fasta(os.path.join(GOLDEN_DIR, "BBA0184.aln"), gapped)
    for s, g in zip(seqs, aligned):
        v = g.get_track(ct.TRACK_ID_INPUT).values
        assert np.array_equal(v[v != 99], s.get_track(ct.TRACK_ID_INPUT).values)


def test_merge_order_incremental_equals_full_recomputation():
    """merge_order keeps the linkage table (and per-row minima) across rounds; the reference recomputes the
    whole table every round (cluster.py:27-114).  The orders must be identical - also on tie-heavy integer
    matrices, where the first-minimum rule decides, and on asymmetric ones."""
    def full(distance_matrix, linkage):
        d = np.asarray(distance_matrix, dtype=float)
        reduce_fn = {'single': np.min, 'complete': np.max, 'average': np.mean}[linkage]
        clusters = {i: [i] for i in range(d.shape[0])}
        order = []
        while len(clusters) > 1:
            ids = list(clusters.keys())
            a = np.full((len(ids), len(ids)), np.inf)
            for i, ci in enumerate(ids):
                for j, cj in enumerate(ids):
                    if ci != cj:
                        a[i, j] = reduce_fn(d[np.ix_(clusters[ci], clusters[cj])])
            i, j = np.unravel_index(a.argmin(), a.shape)
            clusters[ids[i]] = clusters[ids[i]] + clusters[ids[j]]
            del clusters[ids[j]]
            order.append((ids[i], ids[j]))
        return order
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 7, 24, 48):
        for ties in (True, False):
            for symmetric in (True, False):   # the reference evaluates (i, j) and (j, i) separately
                m = rng.integers(0, 6, (n, n)).astype(np.float32) if ties else rng.random((n, n)).astype(np.float32)
                if symmetric:
                    m = m + m.T
                np.fill_diagonal(m, 0)
                for linkage in ('single', 'complete', 'average'):
                    assert comp.merge_order(m, linkage) == full(m, linkage), (n, ties, symmetric, linkage)


def _spy_aligner_class(calls, bba_S):
    """A user-registered aligner component (its own type id): records every request and aligns with the CPU oracle."""
    from oracle import oracle as orc

    class SpyAligner(core.Component):
        tid = "tests.SpyAligner"
        inputs = comp.PairwiseAligner.inputs
        outputs = comp.PairwiseAligner.outputs
        options = {'gap_series': [float], 'debug': int}
        defaults = {'gap_series': [-11.0, -1.0], 'debug': 0}

        def execute(self, mode, sequence_one, sequence_two, track_id_sets_one, track_id_sets_two, zero_idxs, score_matrices):
            gs = list(self.environment['gap_series'])
            calls.append((mode, sequence_one.name, sequence_two.name, tuple(gs), len(zero_idxs or [])))
            p1 = comp._track_profile(sequence_one.get_track(track_id_sets_one[0][0]))
            p2 = comp._track_profile(sequence_two.get_track(track_id_sets_two[0][0]))
            score, path = orc.pairwise_align(mode, [p1], [p2], [score_matrices[0].matrix], gs, zero_idxs=zero_idxs or None)
            yield core.CompleteMessage(outputs={'alignment': ct.Alignment([sequence_one, sequence_two],
                                                                         comp._path_for_output(mode, path)),
                                                'score': float(score)})
    return SpyAligner


def test_callers_honour_the_aligner_seam(bba):
    """GuideTreeBuilder and both master-slave aligners resolve the `aligner` option and collapse `aligner_env` like the
    reference (tree.py:115-127, preprofile.py:127-139,229-241): a user-registered aligner sees EVERY alignment, in the
    reference's order, with the overridden gap_series - and since the spy aligns with the oracle, the outputs must equal
    the goldens produced by the real reference pipeline.  No GPU involved: the device path is bypassed entirely."""
    calls = []
    spy = _spy_aligner_class(calls, bba["S"])
    idx = core.TypeIndex()
    idx.autoregister()
    idx.register(spy)
    manager = core.Manager(idx)
    seqs = [ct.Sequence("seq%03d" % (i + 1), [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
            for i, v in enumerate(bba["seqs"])]
    blosum = ct.blosum62()
    T_IN = [[ct.TRACK_ID_INPUT]]

    def run_one(component, keys, **inputs):
        ex = core.Execution(manager, "root")
        ex.add_task(component).environment(core.Environment({}), core.Environment(keys)).inputs(**inputs)
        return core.run(ex)[0]

    d = load_golden("preprofile.npz")
    for key, component, master, keys, per_slave in (("global_m0_", comp.GlobalMasterSlaveAligner, 0, {}, 1),
                                                    ("local_m4_", comp.LocalMasterSlaveAligner, 4, {}, 2),
                                                    ("local_we3_m0_", comp.LocalMasterSlaveAligner, 0,
                                                     {"waterman_eggert_iterations": 3}, 3)):
        del calls[:]
        slaves = [s for k, s in enumerate(seqs) if k != master]
        out = run_one(component, dict(keys, aligner=spy.tid), master_sequence=seqs[master], slave_sequences=slaves,
                      track_id_sets=T_IN, score_matrices=[blosum])
        assert np.array_equal(np.asarray(out['alignment'].path), d[key + "msa_path"]), key
        assert [c[2] for c in calls] == [s.name for s in slaves for _ in range(per_slave)], key
        assert all(c[1] == seqs[master].name and c[3] == (-11.0, -1.0) for c in calls)
        if per_slave > 1:   # the zero_idxs list grows from iteration to iteration (preprofile.py:247-255)
            assert calls[0][4] == 0 and 0 < calls[1][4] and (per_slave < 3 or calls[1][4] < calls[2][4])

    # the guide tree: one task per pair i < j in row-major order, aligner_env overrides the gap series
    del calls[:]
    out = run_one(comp.GuideTreeBuilder, {"aligner": spy.tid, "aligner_env": core.Environment({"gap_series": [-5.0, -2.0]})},
                  sequences=seqs, track_id_sets=T_IN, score_matrices=[blosum])
    assert [(c[1], c[2]) for c in calls] == [(seqs[i].name, seqs[j].name) for i in range(5) for j in range(i + 1, 5)]
    assert all(c[0] == "global" and c[3] == (-5.0, -2.0) for c in calls)
    # ... and an unregistered aligner is an error, not a silent default
    with pytest.raises(core.ComponentError):
        run_one(comp.GuideTreeBuilder, {"aligner": "no.such.Aligner"}, sequences=seqs, track_id_sets=T_IN,
                score_matrices=[blosum])


def test_entry_point_metadata_registers_the_components(tmp_path):
    """setup.py publishes the components under the reference's `praline.type` entry-point group (setup.py:8-20 there);
    TypeIndex.autoregister reads that group like manager.py:72-85."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)], cwd=root)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from importlib import metadata\n"
            "eps = metadata.entry_points().select(group='praline.type')\n"
            "names = sorted(e.name for e in eps)\n"
            "from praline_amd import core, component\n"
            "assert names == sorted(c.__name__ for c in component.COMPONENTS), names\n"
            "assert all(e.load() is getattr(component, e.name) for e in eps)\n"
            "idx = core.TypeIndex(); idx._types = {}\n"
            "idx.autoregister()\n"
            "assert idx.resolve('praline.component.PairwiseAligner') is component.PairwiseAligner\n"
            "assert len(idx._types) == 9\n" % (str(tmp_path), root))
    subprocess.check_call([sys.executable, "-c", code])


def test_autoregister_is_not_displaced_by_a_second_distribution(tmp_path):
    """The `praline.type` group is shared with the reference distribution, which publishes the SAME tids
    (/root/reference/setup.py:8-20).  With both installed, this package's managers must still resolve its own classes:
    a foreign class under a colliding tid, a class that is not a subclass of this runtime's Component and an entry point
    whose import fails are all skipped; a third-party aligner written against this runtime (new tid) is added."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)], cwd=root)
    other = tmp_path / "other"
    (other / "fakepraline").mkdir(parents=True)
    (other / "fakepraline" / "__init__.py").write_text(
        "from praline_amd import core\n"
        "class Foreign(object):\n"
        "    tid = 'praline.component.PairwiseAligner'\n"
        "class Collides(core.Component):\n"
        "    tid = 'praline.component.GuideTreeBuilder'\n"
        "class MyAligner(core.Component):\n"
        "    tid = 'third.party.MyAligner'\n")
    info = other / "fakepraline-1.0.dist-info"
    info.mkdir()
    (info / "METADATA").write_text("Metadata-Version: 2.1\nName: fakepraline\nVersion: 1.0\n")
    (info / "entry_points.txt").write_text(
        "[praline.type]\n"
        "PairwiseAligner = fakepraline:Foreign\n"
        "GuideTreeBuilder = fakepraline:Collides\n"
        "MyAligner = fakepraline:MyAligner\n"
        "Broken = fakepraline_does_not_exist:Nothing\n")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from praline_amd import core, component\n"
            "import fakepraline\n"
            "import warnings\n"
            "idx = core.TypeIndex()\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    idx.autoregister()\n"
            "for cls in component.COMPONENTS:\n"
            "    assert idx.resolve(cls.tid) is cls, cls.tid\n"
            "assert idx.resolve('third.party.MyAligner') is fakepraline.MyAligner\n"
            "assert len(idx._types) == len(component.COMPONENTS) + 1\n"
            "# nothing is dropped silently: the failed import, the foreign class and the colliding tid are all reported\n"
            "msgs = ' | '.join(str(x.message) for x in w)\n"
            "assert 'Broken' in msgs and 'could not be loaded' in msgs, msgs\n"
            "assert 'not a praline_amd.core.Component subclass' in msgs, msgs\n"
            "assert 'praline.component.GuideTreeBuilder' in msgs and 'is kept' in msgs, msgs\n"
            "# strict: load errors propagate as in the reference (manager.py:72-85)\n"
            "try:\n"
            "    core.TypeIndex().autoregister(strict=True)\n"
            "except ImportError:\n"
            "    pass\n"
            "else:\n"
            "    raise AssertionError('strict autoregister swallowed a load error')\n" % (str(other), str(tmp_path), root))
    subprocess.check_call([sys.executable, "-c", code])


def test_native_merge_order_equals_numpy_version():
    """praline_merge_order (csrc/cluster.cpp, host code of the library) against component.merge_order's numpy statement
    of the same incremental algorithm: every linkage, tie-heavy integer distances (integer scoring), float distances,
    asymmetric matrices; and the guide-tree golden of the real reference."""
    from praline_amd import native
    rng = np.random.default_rng(5)
    saved = comp.native_clustering
    try:
        for n in (2, 3, 17, 64, 130, 257):
            for kind in ("ties", "float", "asym"):
                m = rng.integers(0, 9, (n, n)).astype(np.float64) if kind == "ties" else rng.random((n, n)) * 100
                if kind != "asym":
                    m = m + m.T
                np.fill_diagonal(m, 0)
                m = m.astype(np.float32)        # the guide tree hands over float32 distances (tree.py:147)
                for linkage in ("single", "complete", "average"):
                    comp.native_clustering = False
                    want = comp.merge_order(m, linkage)
                    assert native.merge_order(m, linkage) == want, (n, kind, linkage)
                    comp.native_clustering = True
                    assert comp.merge_order(m, linkage) == want
    finally:
        comp.native_clustering = saved
    d = load_golden("profile_profile.npz")
    assert native.merge_order(d["dist"], "average") == [tuple(x) for x in d["merge_order"]]


def test_merge_levels_groups_independent_steps():
    """component.merge_levels: the merge steps of a guide tree grouped into levels of mutually independent steps (the
    resident TreeMSA path submits one batch per level): every step comes after the steps that produced its clusters, no
    two steps of a level share a cluster, a caterpillar tree degenerates to one step per level."""
    from praline_amd.component import merge_levels
    assert merge_levels([(0, 1), (0, 2), (0, 3), (0, 4)]) == [[0], [1], [2], [3]]
    assert merge_levels([(2, 3), (0, 1), (4, 5), (0, 2), (6, 7), (4, 6), (0, 4)]) == [[0, 1, 2, 4], [3, 5], [6]]
    rng = np.random.default_rng(3)
    for n in (2, 3, 17, 100):
        alive = list(range(n))
        steps = []
        while len(alive) > 1:
            a, b = sorted(rng.choice(len(alive), 2, replace=False))
            steps.append((alive[a], alive[b]))
            del alive[b]
        levels = merge_levels(steps)
        assert sorted(k for lv in levels for k in lv) == list(range(n - 1))
        level_of = {k: q for q, lv in enumerate(levels) for k in lv}
        made = {}                                  # cluster -> step that last produced it
        for k, (i, j) in enumerate(steps):
            for c in (i, j):
                if c in made:
                    assert level_of[made[c]] < level_of[k]
            made[i] = k
        for lv in levels:
            used = [c for k in lv for c in steps[k]]
            assert len(used) == len(set(used))


def test_load_score_matrix_reads_the_reference_text_format(tmp_path):
    """io.load_score_matrix / io.open_builtin (praline/__init__.py:57-102): every table the reference packages
    (praline/matrices/*: 15 BLOSUM tables + nucleotide) parses to the array the reference's own loader produces
    (tests/golden/matrices.npz, written by make_golden.py from the real reference); comments, ragged rows, extra columns
    and the anonymous alphabet follow the reference."""
    from praline_amd import io as pio
    from praline_amd import matrices
    want = load_golden("matrices.npz")
    assert sorted(want.files) == matrices.builtin_names()
    for name in want.files:
        alphabet = ct.ALPHABET_DNA if name == "nucleotide" else ct.ALPHABET_AA
        m = pio.load_score_matrix(pio.open_builtin("matrices/" + name), alphabet=alphabet)
        assert m.matrix.dtype == np.float32 and np.array_equal(m.matrix, want[name]), name
        assert m.alphabets == [alphabet, alphabet]
    assert np.array_equal(pio.load_score_matrix(pio.open_builtin("matrices/blosum62"), ct.ALPHABET_AA).matrix, ct.blosum62().matrix)
    with pytest.raises(core.DataError):
        pio.open_builtin("matrices/blosum63")
    # a file on disk: comment lines, a trailing comment, a value beyond the header (ignored), no alphabet given
    path = tmp_path / "m.txt"
    path.write_text("# toy\n   *  M   # columns\n*  0  0  99\nM  0  15\n")
    m = pio.load_score_matrix(str(path))
    assert m.alphabets[0] is m.alphabets[1] and m.alphabets[0].aid.startswith("__anonymous_from_matrix_")
    assert m.alphabets[0].symbol_to_index("*") == 0 and m.alphabets[0].symbol_to_index("M") == 1 and m.alphabets[0].size == 2
    assert np.array_equal(m.matrix, np.array([[0, 0], [0, 15]], dtype=np.float32))
    # the shipped annotation matrix (extra/data/motif_score_matrix), from bytes
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "motif_score_matrix"), "rb") as f:
        m2 = pio.load_score_matrix(f)
    assert np.array_equal(m2.matrix, m.matrix)
    with pytest.raises(core.AlphabetError):
        pio.load_score_matrix(pio.open_builtin("matrices/blosum62"), alphabet=ct.ALPHABET_DNA)   # symbols the alphabet lacks
