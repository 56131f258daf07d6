"""All-pairs distance stage: pair enumeration and sharding.

Mirrors the pair loop of GuideTreeBuilder.execute (praline/component/tree.py:105-131): pairs
(i, j), i < j, row-major; scores are scattered into the symmetric matrix d and the distance is
(-d) + d.max() with the zero diagonal included in the max (tree.py:142-147)."""
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
