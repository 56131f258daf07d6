"""world_size-2 test of the sharded all-pairs path on CPU (gloo): sharding, the all-gather of the
score slices and the reassembly.  The GPU scorer is replaced by the CPU oracle here (this test
checks the host / collective logic, not the kernels)."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import load_golden, one_hot
    from oracle import oracle as orc
    from praline_amd import allpairs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = load_golden("synthetic_c1.npz")
    S = load_golden("bba0184_inputs.npz")["blosum62"]
    profs = [one_hot(d["seq%d" % i], 27) for i in range(8)]
    calls = []

    def scorer(pairs):
        calls.append(len(pairs))
        return torch.tensor([orc.pairwise_score_fast("global", profs[i], profs[j], S, -11.0, -1.0)
                             for i, j in pairs], dtype=torch.float32)

    pairs, scores = allpairs.all_pairs_scores([p.shape[0] for p in profs], scorer, rank, world)
    np.save(os.path.join(out_dir, "scores_%d.npy" % rank), scores.numpy())
    np.save(os.path.join(out_dir, "ncalls_%d.npy" % rank), np.array(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_all_pairs_world2(tmp_path):
    import torch.multiprocessing as mp
    from conftest import load_golden
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = load_golden("synthetic_c1.npz")["scores_global"]
    s0 = np.load(tmp_path / "scores_0.npy")
    s1 = np.load(tmp_path / "scores_1.npy")
    assert np.array_equal(s0, s1)                       # every rank holds the full score list
    assert np.array_equal(s0.astype(np.float64), ref)   # and it is the reference's
    n0, n1 = int(np.load(tmp_path / "ncalls_0.npy")[0]), int(np.load(tmp_path / "ncalls_1.npy")[0])
    assert n0 + n1 == 28 and 0 < n0 < 28                # each rank aligned only its slice


# ---- operator level: GuideTreeBuilder under BatchManager(rank, world) and the master-sharded preprofile stage ----
def _cpu_seams(comp, ct, util, orc, np):
    """CPU stand-ins for the two DEVICE steps (the tests below check sharding, exchange and reassembly, not kernels)."""

    def scores_for_pairs(self, sequences, ii, jj, modes):
        out = np.zeros(len(ii), dtype=np.float32)
        for k, (i, j, mode) in enumerate(zip(ii, jj, modes)):
            p1 = comp._track_profile(sequences[i].get_track(self.ids_one[0]))
            p2 = comp._track_profile(sequences[j].get_track(self.ids_one[0]))
            out[k] = orc.pairwise_score_fast(str(mode), p1, p2, self.S, self.gap_open, self.gap_extend)
        return out

    def slave_counts(profiles, S, pairs, mode, gap_open, gap_extend, score_threshold, iterations, counts_out=None):
        lens = np.array([p.shape[0] for p in profiles], dtype=np.int32)
        row_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
        cat = np.concatenate(profiles, axis=0)
        seqs = [ct.Sequence("s%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=p.argmax(axis=1)))])
                for i, p in enumerate(profiles)]
        results = [[] for _ in pairs]
        rects = [[] for _ in pairs]
        for _ in range(iterations if len(pairs) else 0):
            sc, paths = orc.batch_align([mode], cat, row_off, lens, S, pairs, gap_open, gap_extend, rects=rects)
            for k in range(len(pairs)):
                p = np.array(paths[k][0], dtype=int)
                results[k].append((float(sc[k, 0]), p))
                rects[k].append((int(p[:, 0].min()), int(p[:, 0].max()), int(p[:, 1].min()), int(p[:, 1].max())))
        counts = np.zeros((int(lens.sum()), S.shape[0]), dtype=np.int32)
        for m in sorted(set(int(a) for a, _ in pairs)):
            ks = [k for k in range(len(pairs)) if pairs[k][0] == m]
            aln = comp.merge_master_slave(seqs[m], [seqs[int(pairs[k][1])] for k in ks], [results[k] for k in ks],
                                          score_threshold, local=(mode == "local"))
            freqs = np.array(util.get_frequencies(aln, ct.TRACK_ID_INPUT), dtype=np.int32)
            freqs[np.arange(lens[m]), profiles[m].argmax(axis=1)] -= 1       # the master's own symbols: added by the caller
            counts[row_off[m]:row_off[m] + lens[m]] = freqs
        return counts

    return scores_for_pairs, slave_counts


def _operator_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from conftest import load_golden
    from oracle import oracle as orc
    from praline_amd import component as comp
    from praline_amd import container as ct
    from praline_amd import core, util

    group = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scores_for_pairs, slave_counts = _cpu_seams(comp, ct, util, orc, np)
    calls = {"pairs": 0, "prepairs": 0}

    def counted_scores(self, sequences, ii, jj, modes, on_device=False):
        assert not on_device            # (gloo: the shard is a numpy array; under RCCL it stays on the device)
        calls["pairs"] += len(ii)
        return scores_for_pairs(self, sequences, ii, jj, modes)

    def counted_counts(profiles, S, pairs, *args, **kwargs):
        calls["prepairs"] += len(pairs)
        return slave_counts(profiles, S, pairs, *args, **kwargs)

    saved = (comp.PairwiseBatch.scores_for_pairs, comp._preprofile_slave_counts)
    comp.PairwiseBatch.scores_for_pairs = counted_scores
    comp._preprofile_slave_counts = counted_counts
    try:
        _operator_body(rank, world, group, out_dir, comp, ct, core, calls, load_golden)
    finally:
        comp.PairwiseBatch.scores_for_pairs, comp._preprofile_slave_counts = saved
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _operator_body(rank, world, group, out_dir, comp, ct, core, calls, load_golden):
    d = load_golden("synthetic_c1.npz")
    seqs = [ct.Sequence("seq%d" % i, [(ct.TRACK_ID_INPUT, ct.PlainTrack(None, ct.ALPHABET_AA, raw_indices=d["seq%d" % i]))])
            for i in range(8)]
    blosum = ct.blosum62()
    idx = core.TypeIndex()
    idx.autoregister()
    manager = comp.BatchManager(idx, rank=rank, world=world, group=group)
    out = {}
    for dist_mode in ("global", "semiglobal_auto"):
        ex = core.Execution(manager, "root")
        ex.add_task(comp.GuideTreeBuilder).environment(core.Environment({}), core.Environment({"dist_mode": dist_mode})).inputs(
            sequences=seqs, track_id_sets=[[ct.TRACK_ID_INPUT]], score_matrices=[blosum])
        tree = core.run(ex)[0]['guide_tree']
        out["merge_" + dist_mode] = np.array(tree.merge_orders)
    out["tree_pairs"] = np.array(calls["pairs"])
    for mode, it in (("global", 1), ("local", 2)):
        tracks = comp.build_preprofiles(seqs, ct.TRACK_ID_INPUT, blosum, mode=mode, waterman_eggert_iterations=it,
                                        score_threshold=20.0 if mode == "local" else None, rank=rank, world=world, group=group)
        out["counts_" + mode] = np.concatenate([np.asarray(t.counts) for t in tracks], axis=0)
    out["pre_pairs"] = np.array(calls["prepairs"])
    np.savez(os.path.join(out_dir, "op_w%d_r%d.npz" % (world, rank)), **out)


def test_operator_multi_rank_equals_single_rank(tmp_path):
    """GuideTreeBuilder under BatchManager(rank, world) and build_preprofiles(rank, world) at world 2 and 3 (gloo):
    every rank ends with the single-rank merge orders / count tracks bit for bit, and aligned only its share."""
    import torch.multiprocessing as mp
    _operator_worker(0, 1, 0, str(tmp_path))
    single = np.load(tmp_path / "op_w1_r0.npz")
    assert int(single["tree_pairs"]) == 2 * 28 and int(single["pre_pairs"]) == 2 * 56
    for world in (2, 3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_operator_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        tree_pairs = pre_pairs = 0
        for r in range(world):
            got = np.load(tmp_path / ("op_w%d_r%d.npz" % (world, r)))
            for key in ("merge_global", "merge_semiglobal_auto", "counts_global", "counts_local"):
                assert np.array_equal(got[key], single[key]), (world, r, key)
            assert 0 < int(got["tree_pairs"]) < int(single["tree_pairs"])
            assert 0 < int(got["pre_pairs"]) < int(single["pre_pairs"])
            tree_pairs += int(got["tree_pairs"])
            pre_pairs += int(got["pre_pairs"])
        assert tree_pairs == int(single["tree_pairs"]) and pre_pairs == int(single["pre_pairs"])


def test_shard_masters_balance():
    from praline_amd import allpairs
    rng = np.random.default_rng(0)
    lens = rng.integers(100, 400, 1024)
    for world in (1, 2, 3, 8):
        shards = allpairs.shard_masters(lens, world)
        allm = np.sort(np.concatenate(shards))
        assert np.array_equal(allm, np.arange(1024))
        load = [int((lens[s] * (lens.sum() - lens[s])).sum()) for s in shards]
        assert max(load) <= 1.01 * (sum(load) / world)
