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
