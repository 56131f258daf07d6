"""All-pairs distance stage: pair enumeration, sharding over ranks and the score exchange.

Mirrors the pair loop of GuideTreeBuilder.execute (praline/component/tree.py:105-131): pairs
(i, j), i < j, row-major; scores are scattered into the symmetric matrix d and the distance is
(-d) + d.max() with the zero diagonal included in the max (tree.py:142-147).

Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The pair list
is cut into contiguous, cell-balanced slices; every rank aligns its slice with the HIP path and the
score slices are exchanged with ONE all-gather (the path has no other data exchange).  The profile
arena is replicated on every GPU (it is tiny next to 288 GB of HBM).
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
    import torch
    from . import native

    arena = native.Arena(profiles, score_matrix)

    def score(pairs):
        out = torch.zeros(len(pairs), dtype=torch.float32, device="cuda")
        if len(pairs):
            plan = native.Plan(arena, pairs)
            try:
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
    cells = lens[pairs[:, 0]] * lens[pairs[:, 1]]
    bounds = shard_bounds(cells, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = scorer(pairs[lo:hi])
    if world == 1:
        return pairs, mine
    import torch.distributed as dist
    slice_len = max(bounds[r + 1] - bounds[r] for r in range(world))
    padded = torch.zeros(slice_len, dtype=torch.float32, device=mine.device if device is None else device)
    padded[:hi - lo] = mine
    gathered = torch.zeros(slice_len * world, dtype=torch.float32, device=padded.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)   # the one exchange step of the path
    out = torch.cat([gathered[r * slice_len:r * slice_len + (bounds[r + 1] - bounds[r])] for r in range(world)])
    return pairs, out


def guide_tree_distance(profiles, score_matrix, mode="global", gap_open=-11.0, gap_extend=-1.0, rank=0, world=1):
    """d and dist = (-d) + d.max() of GuideTreeBuilder (tree.py:99-147) for raw profiles."""
    scorer = device_scorer(profiles, score_matrix, mode, gap_open, gap_extend)
    pairs, scores = all_pairs_scores([p.shape[0] for p in profiles], scorer, rank, world)
    return scores_to_distance(len(profiles), pairs, scores.cpu().numpy())
