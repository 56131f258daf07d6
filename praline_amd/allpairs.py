"""All-pairs distance stage: pair enumeration, sharding over ranks and the score exchange.

Mirrors the pair loop of GuideTreeBuilder.execute (praline/component/tree.py:105-131): pairs
(i, j), i < j, row-major; scores are scattered into the symmetric matrix d and the distance is
(-d) + d.max() with the zero diagonal included in the max (tree.py:142-147).

Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The pair list
is sharded by COLUMNS: all pairs (i, j) of one sequence two = j go to the same rank, columns dealt to
ranks balanced by DP cells.  The device path packs 32 pairs that share sequence two into one wavefront
task, so whole columns keep the tasks as full as on one GPU (contiguous row-major slices would leave a
rank with few i per j: up to 25 % more, emptier tasks at 8 ranks).  Every rank aligns its shard with
the HIP path and the score shards are exchanged with ONE all-gather (the path has no other data
exchange), then put back into the reference's pair order.  The profile arena is replicated on every
GPU (it is tiny next to 288 GB of HBM).
"""
import numpy as np


def enumerate_pairs(n):
    """(i, j) for i < j in the reference's order (tree.py:105-129)."""
    i, j = np.triu_indices(n, k=1)
    return np.stack([i, j], axis=1).astype(np.int32)


def shard_bounds(cells, world):
    """Contiguous pair-index ranges balanced by the DP cell count (sum L1*L2): bounds[r]..bounds[r+1]
    is rank r's slice.  Deterministic and identical on every rank."""
    cells = np.asarray(cells, dtype=np.int64)
    total = int(cells.sum())
    csum = np.cumsum(cells)
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(csum, total * r / world, side="left")))
    bounds.append(len(cells))
    for r in range(1, len(bounds)):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds


def shard_columns(lens, pairs, world):
    """Per rank, the (ascending) indices into the row-major pair list of the pairs it aligns: whole
    columns j, dealt longest-first to the least loaded rank (load = DP cells).  Deterministic and
    identical on every rank."""
    lens = np.asarray(lens, dtype=np.int64)
    pairs = np.asarray(pairs)
    cells = lens[pairs[:, 0]] * lens[pairs[:, 1]]
    n = len(lens)
    col_cells = np.zeros(n, dtype=np.int64)
    np.add.at(col_cells, pairs[:, 1], cells)
    owner = np.zeros(n, dtype=np.int64)
    load = np.zeros(world, dtype=np.int64)
    for j in np.argsort(-col_cells, kind="stable"):
        r = int(np.argmin(load))
        owner[j] = r
        load[r] += col_cells[j]
    pair_owner = owner[pairs[:, 1]]
    return [np.nonzero(pair_owner == r)[0].astype(np.int64) for r in range(world)]


def gather_maps(shards):
    """Index maps of the exchange step: every rank contributes a slice padded to the longest shard, the
    all-gather lines the slices up rank by rank, and `out[dst] = gathered[src]` puts the scores back into the
    row-major pair order.  Returns (src, dst, shard_len)."""
    shard_len = int(max(len(ix) for ix in shards))
    src = np.concatenate([r * shard_len + np.arange(len(ix), dtype=np.int64) for r, ix in enumerate(shards)])
    dst = np.concatenate([np.asarray(ix, dtype=np.int64) for ix in shards])
    return src, dst, shard_len


def shard_masters(lens, world):
    """Preprofile stage ("shard by master index"): every master i is aligned against all other
    sequences, DP cells = L_i * (sum L - L_i).  Masters are dealt longest-first to the least loaded rank, so all
    N - 1 alignments of a master - and with them its whole count block - live on ONE rank.  Returns per rank
    the ascending master indices.  Deterministic and identical on every rank."""
    lens = np.asarray(lens, dtype=np.int64)
    cells = lens * (lens.sum() - lens)
    owner = np.zeros(len(lens), dtype=np.int64)
    load = np.zeros(world, dtype=np.int64)
    for i in np.argsort(-cells, kind="stable"):
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cells[i]
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def _require_torch_gpu():
    """torch must have been imported BEFORE libpraline_dp.so was loaded (it ships its own HIP runtime; loaded second, it
    reports no GPU): fail with that explanation rather than with torch's."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("torch sees no GPU - import torch before praline_amd.native loads libpraline_dp.so "
                           "(torch.distributed programs do: the process group comes first)")
    return torch


def _group_device(group):
    """Tensors of a collective live where the group's backend wants them: cuda for nccl (= RCCL), cpu for gloo."""
    import torch
    import torch.distributed as dist
    if dist.get_backend(group) != "nccl":
        return torch.device("cpu")
    _require_torch_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def group_is_nccl(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"


def all_gather_scores(local, shards, rank, world, group=None):
    """The exchange step of the all-pairs stage for host-side callers (GuideTreeBuilder): `local` = this rank's
    scores in the order of shards[rank] - a numpy array, or a torch tensor that is already on the group's device (the
    kernels wrote into it: nothing moves through host memory before the exchange); returns the complete float32 list in
    pair order as a numpy array, identical on every rank.  One all_gather_into_tensor (RCCL over xGMI under the nccl
    backend), the reorder into pair order on the same device, one copy to the host."""
    import torch
    import torch.distributed as dist
    src, dst, shard_len = gather_maps(shards)
    dev = _group_device(group)
    padded = torch.zeros(shard_len, dtype=torch.float32, device=dev)
    if isinstance(local, torch.Tensor):
        padded[:len(shards[rank])] = local.to(dev)
    else:
        padded[:len(shards[rank])] = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float32), device=dev)
    gathered = torch.zeros(shard_len * world, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    ordered = torch.zeros(sum(len(ix) for ix in shards), dtype=torch.float32, device=dev)
    ordered[torch.as_tensor(dst, device=dev)] = gathered[torch.as_tensor(src, device=dev)]
    return ordered.cpu().numpy()


def all_reduce_counts(counts, group=None):
    """The exchange step of the preprofile stage: every rank holds the count blocks of ITS masters (zeros elsewhere);
    one all-reduce (sum) of the int32 count arena [sum L][A] gives every rank all of them.  counts: a torch tensor
    (on the group's device; reduced in place) or a numpy array (returned reduced)."""
    import torch
    import torch.distributed as dist
    if isinstance(counts, np.ndarray):
        t = torch.as_tensor(np.ascontiguousarray(counts, dtype=np.int32), device=_group_device(group))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def scores_to_distance(n, pairs, scores):
    """tree.py:99-100,131,142-147: d filled symmetrically, diagonal 0, dist = (-d) + d.max()."""
    d = np.zeros((n, n), dtype=np.float32)
    pairs = np.asarray(pairs)
    d[pairs[:, 0], pairs[:, 1]] = scores
    d[pairs[:, 1], pairs[:, 0]] = scores
    return d, (-d) + d.max()


def device_scorer(profiles, score_matrix, mode, gap_open, gap_extend):
    """The product scorer: align a slice of the pair list on this rank's GPU (scores stay in HBM).
    Returns a function pairs -> torch.cuda.FloatTensor."""
    torch = _require_torch_gpu()
    from . import native

    arena = native.Arena(profiles, score_matrix)

    def score(pairs):
        # every entry is written by the DP kernel: no fill.  The kernel runs on the LIBRARY's stream, which knows
        # nothing of torch's: order it behind whatever torch has queued on its current stream (a recycled
        # caching-allocator block may still be in use there), and finish it before torch sees `out`.
        out = torch.empty(len(pairs), dtype=torch.float32, device="cuda")
        if len(pairs):
            plan = native.Plan(arena, pairs)
            try:
                torch.cuda.ExternalStream(native.stream_handle()).wait_stream(torch.cuda.current_stream())
                plan.run(mode, gap_open, gap_extend, d_scores=out.data_ptr())
                native.synchronize()
            finally:
                plan.close()
        return out

    score.arena = arena
    return score


def all_pairs_scores(lens, scorer, rank=0, world=1, group=None, device=None):
    """Scores of all pairs i < j, identical on every rank.

    lens: sequence lengths (for cell balancing); scorer(pairs) -> 1-D float32 tensor for a slice of
    the pair list (device_scorer on a GPU; the gloo tests inject a CPU scorer); group: torch
    process group (None = default group) when world > 1."""
    import torch
    lens = np.asarray(lens, dtype=np.int64)
    pairs = enumerate_pairs(len(lens))
    if world == 1:
        return pairs, scorer(pairs)
    import torch.distributed as dist
    shards = shard_columns(lens, pairs, world)
    mine = scorer(pairs[shards[rank]])
    src, dst, shard_len = gather_maps(shards)
    dev = mine.device if device is None else device
    padded = torch.zeros(shard_len, dtype=torch.float32, device=dev)
    padded[:len(shards[rank])] = mine
    gathered = torch.empty(shard_len * world, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)   # the one exchange step of the path
    out = torch.zeros(len(pairs), dtype=torch.float32, device=dev)  # back into the reference's pair order
    out[torch.as_tensor(dst, device=dev)] = gathered[torch.as_tensor(src, device=dev)]
    return pairs, out


def guide_tree_distance(profiles, score_matrix, mode="global", gap_open=-11.0, gap_extend=-1.0, rank=0, world=1):
    """d and dist = (-d) + d.max() of GuideTreeBuilder (tree.py:99-147) for raw profiles."""
    scorer = device_scorer(profiles, score_matrix, mode, gap_open, gap_extend)
    pairs, scores = all_pairs_scores([p.shape[0] for p in profiles], scorer, rank, world)
    return scores_to_distance(len(profiles), pairs, scores.cpu().numpy())
