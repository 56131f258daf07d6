"""Error behaviour of the C ABI and the operator layer on a GPU box: misuse comes back as an error code with a
message (the reference has undefined behaviour there, praline/component/align.py:196-199), component-level
mistakes raise the reference's exception types."""
import numpy as np
import pytest

from conftest import one_hot
from praline_amd import component as comp, container as ct, core

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from praline_amd import native
    native.init(0)
    return native


def test_c_abi_rejects_misuse(nat, bba):
    S = bba["S"]
    p = [one_hot([1, 2, 3, 4], 27), one_hot([4, 3, 2], 27)]
    with pytest.raises(ValueError):
        nat.Arena(p, np.zeros((5, 5), np.float32))                       # profile width != matrix size
    with pytest.raises(nat.NativeError) as e:
        nat.Arena([np.zeros((0, 27), np.float32)], S)                    # empty sequence
    assert e.value.code == -1 and "length" in str(e.value)
    with pytest.raises(nat.NativeError) as e:                            # alphabet wider than the ABI's 254
        nat.Arena([np.eye(255, dtype=np.float32)], np.ones((255, 255), np.float32))
    assert e.value.code == -1
    arena = nat.Arena(p, S)
    with pytest.raises(nat.NativeError) as e:
        nat.Plan(arena, np.array([(0, 2)], np.int32))                    # sequence index out of range
    assert e.value.code == -1
    plan = nat.Plan(arena, np.array([(0, 1)], np.int32))
    with pytest.raises(KeyError):
        plan.run("diagonal", -11, -1)                                    # unknown mode (binding)
    plan.run("global", -11, -1)
    with pytest.raises(nat.NativeError) as e:
        plan.paths()                                                     # plan was created without paths
    assert "want_paths" in str(e.value)
    with pytest.raises(nat.NativeError):
        plan.add_counts()
    plan.close()
    soft = nat.Arena([np.full((4, 27), 1.0 / 27, np.float32), p[1]], S)  # a non-one-hot profile
    soft.counts_reset()
    plan = nat.Plan(soft, np.array([(0, 1)], np.int32), want_paths=True)
    plan.run("global", -11, -1)
    with pytest.raises(nat.NativeError) as e:
        plan.add_counts()                                                # counting needs plain sequences
    assert e.value.code == -4 and "one-hot" in str(e.value)
    plan.close()
    soft.close()
    with pytest.raises(nat.NativeError):
        nat.Plan(arena, np.array([(0, 1)], np.int32), want_paths=False, rects=[[(1, 1, 1, 1)]])   # masks need paths
    arena.close()
    # per-position gap scores and a growing arena exclude each other, in both orders (the gap rows exist for the
    # sequences present when they were set: an appended sequence would read zeros or past the buffer)
    gaps = [np.tile(np.float32([-11, -1]), (len(q), 1)) for q in p]
    counts = np.concatenate(p).astype(np.int32)
    grow = nat.Arena(p, S)
    grow.set_counts(counts)                                              # no reserve: cap_seqs stays 0
    with pytest.raises(nat.NativeError) as e:
        grow.set_gap_scores(gaps)
    assert e.value.code == -4 and "growing" in str(e.value)
    grow.close()
    gapped = nat.Arena(p, S)
    gapped.set_gap_scores(gaps)
    with pytest.raises(nat.NativeError) as e:
        gapped.set_counts(counts, reserve_seqs=2, reserve_rows=16)
    assert e.value.code == -4 and "cannot grow" in str(e.value)
    gapped.set_gap_scores(None)                                          # removing them makes the arena growable again
    gapped.set_counts(counts, reserve_seqs=2, reserve_rows=16)
    gapped.close()


def test_component_errors_match_reference_types(nat, bba):
    idx = core.TypeIndex()
    idx.autoregister()
    manager = core.Manager(idx)
    blosum = ct.blosum62()
    mk = lambda name, v: ct.Sequence(name, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=v))])
    a, b = mk("a", [1, 2, 3, 4, 5]), mk("b", [5, 4, 3])

    def run(component, keys=None, **inputs):
        ex = core.Execution(manager, "root")
        ex.add_task(component).environment(core.Environment({}), core.Environment(dict(keys or {}))).inputs(**inputs)
        return core.run(ex)[0]
    T = [[ct.TRACK_ID_INPUT]]
    with pytest.raises(core.ComponentError):                             # align.py:114-117
        run(comp.PairwiseAligner, mode="sideways", sequence_one=a, sequence_two=b, track_id_sets_one=T,
            track_id_sets_two=T, score_matrices=[blosum])
    with pytest.raises(core.ComponentError):                             # align.py:124-127: one track id per set
        run(comp.PairwiseAligner, mode="global", sequence_one=a, sequence_two=b,
            track_id_sets_one=[[ct.TRACK_ID_INPUT, ct.TRACK_ID_INPUT]], track_id_sets_two=T, score_matrices=[blosum])
    with pytest.raises(core.ComponentError):                             # align.py:186-189: gap series of > 2 values
        run(comp.PairwiseAligner, {"gap_series": [-11.0, -1.0, -0.5]}, mode="global", sequence_one=a, sequence_two=b,
            track_id_sets_one=T, track_id_sets_two=T, score_matrices=[blosum])
    with pytest.raises(core.ComponentError):                             # msa.py:109-111
        run(comp.TreeMultipleSequenceAligner, {"merge_mode": "local"}, sequences=[a, b],
            guide_tree=ct.SequenceTree([a, b], [(0, 1)]), track_id_sets=T, score_matrices=[blosum])
    # many Waterman-Eggert iterations (more masked rectangles per pair than the split-strip kernels hold): still one
    # device submission per iteration, and the device preprofile stage agrees with the component chain
    out = run(comp.LocalMasterSlaveAligner, {"waterman_eggert_iterations": 7}, master_sequence=a, slave_sequences=[b],
              track_id_sets=T, score_matrices=[blosum])
    assert np.asarray(out['alignment'].path).shape == (6, 8)
    tracks = comp.build_preprofiles([a, b], ct.TRACK_ID_INPUT, blosum, mode="local", waterman_eggert_iterations=7)
    prof = run(comp.ProfileBuilder, alignment=out['alignment'], track_id=ct.TRACK_ID_INPUT)
    assert np.array_equal(np.asarray(tracks[0].counts), np.asarray(prof['profile_track'].counts))


def test_arena_inputs_staged_or_not_give_the_same_arena(nat, bba):
    """native.Arena concatenates a list of float32 profiles into page-locked staging (praline_host_alloc: the upload is
    a DMA); float64 / mixed lists and lists beyond the staging cap take the pageable road.  Same arena either way -
    scores bit for bit - and the staging buffer is reused (and grown) across arenas."""
    rng = np.random.default_rng(11)
    lens = rng.integers(5, 90, 40)
    profs = []
    for L in lens:
        c = rng.integers(0, 5, (int(L), 27)).astype(np.float32) + np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))]
        profs.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
    pairs = np.array([(i, j) for i in range(40) for j in range(i + 1, 40)], dtype=np.int32)

    def scores(plist):
        a = nat.Arena(plist, bba["S"])
        p = nat.Plan(a, pairs)
        p.run("global", -11.0, -1.0)
        s = p.scores().copy()
        p.close(); a.close()
        return s

    want = scores(profs)                                                   # float32 list: staged
    assert nat._stage["ptr"] is not None
    ptr0 = nat._stage["ptr"].value
    assert np.array_equal(want.view(np.uint32), scores([p.astype(np.float64) for p in profs]).view(np.uint32))   # cast on the way
    mixed = [p.astype(np.float64) if i % 3 == 0 else p for i, p in enumerate(profs)]
    assert np.array_equal(want.view(np.uint32), scores(mixed).view(np.uint32))
    assert np.array_equal(want.view(np.uint32), scores([np.asfortranarray(p) for p in profs]).view(np.uint32))  # any strides
    cap = nat._STAGE_CAP
    try:
        nat._STAGE_CAP = 1024                                              # nothing fits: pageable upload
        assert np.array_equal(want.view(np.uint32), scores(profs).view(np.uint32))
    finally:
        nat._STAGE_CAP = cap
    assert np.array_equal(want.view(np.uint32), scores(profs).view(np.uint32))
    assert nat._stage["ptr"].value == ptr0                                 # the same staging buffer again
    parts_min = nat._PARTS_MIN_BYTES
    try:
        nat._PARTS_MIN_BYTES = 0                                           # begin / put first half / put second half / finish
        assert np.array_equal(want.view(np.uint32), scores(profs).view(np.uint32))
        onehots = [np.eye(27, dtype=np.float32)[rng.integers(0, 20, int(L))] for L in lens]
        got_parts = scores(onehots)
    finally:
        nat._PARTS_MIN_BYTES = parts_min
    assert np.array_equal(got_parts.view(np.uint32), scores(onehots).view(np.uint32))   # (an exact-mode arena both ways)
    big = [np.tile(p, (40, 1)) for p in profs]                             # 40 x the rows: the staging buffer grows
    a = nat.Arena(big + big + big, bba["S"])
    assert int(a.lens.sum()) == 120 * int(lens.sum())
    a.close()
    # the three-step creation, misused
    import ctypes
    L = nat.lib()
    lens32 = np.ascontiguousarray(lens, dtype=np.int32)
    cat = np.ascontiguousarray(np.concatenate(profs, axis=0))
    h = ctypes.c_void_p()
    assert L.praline_arena_begin(len(lens32), lens32.ctypes.data, 27, ctypes.byref(h)) == 0
    assert L.praline_arena_put_rows(h, 5, len(cat), cat.ctypes.data) != 0          # past the end
    assert L.praline_arena_put_rows(h, 0, 10, None) != 0
    hp = ctypes.c_void_p()
    assert L.praline_plan_create(h, len(pairs), pairs.ctypes.data, 0, None, None, ctypes.byref(hp)) != 0   # still being built
    assert L.praline_arena_premultiply(h) != 0
    assert L.praline_arena_put_rows(h, 0, len(cat), cat.ctypes.data) == 0
    S32 = np.ascontiguousarray(bba["S"], dtype=np.float32)
    assert L.praline_arena_finish(h, S32.ctypes.data) == 0
    assert L.praline_arena_finish(h, S32.ctypes.data) != 0                         # not being built any more
    assert L.praline_arena_put_rows(h, 0, 1, cat.ctypes.data) != 0
    assert L.praline_plan_create(h, len(pairs), pairs.ctypes.data, 0, None, None, ctypes.byref(hp)) == 0
    assert L.praline_plan_run(hp, 0, -11.0, -1.0, None) == 0
    got = np.zeros(len(pairs), dtype=np.float32)
    assert L.praline_plan_scores(hp, got.ctypes.data) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    L.praline_plan_destroy(hp); L.praline_arena_destroy(h)
    assert L.praline_arena_begin(len(lens32), lens32.ctypes.data, 27, ctypes.byref(h)) == 0
    assert L.praline_arena_destroy(h) == 0                                        # an arena that was never finished
    assert L.praline_arena_begin(0, lens32.ctypes.data, 27, ctypes.byref(h)) != 0
    # the allocator itself
    p = ctypes.c_void_p()
    assert nat.lib().praline_host_alloc(1 << 20, ctypes.byref(p)) == 0 and p.value
    np.ctypeslib.as_array((ctypes.c_float * 16).from_address(p.value))[:] = 1.0
    assert nat.lib().praline_host_free(p) == 0
    assert nat.lib().praline_host_free(None) == 0
    assert nat.lib().praline_host_alloc(16, None) != 0
