"""TEST INFRASTRUCTURE ONLY - CPU baseline worker: the REAL reference C extension (oracle/_ref, compiled
by oracle/build_ref.sh from the reference's own praline/util/cext.c) timed over a share of a pair list.

Run as a child process by bench.py's `cpu_baseline` leg (one process per host core: the extension holds the
GIL, the reference itself scales by forking processes, praline/core/manager.py).  The process never touches
the GPU and never reads /root/reference.  Per pair it does what PairwiseAligner/RawPairwiseAligner.execute
do around the two C calls (praline/component/align.py:163-221, 357-431), with the Python-level parts
written with numpy so that they cost the reference as little as possible:

    cext_build_scores(P1, P2, nz1, nz2, S, m)      nz = nonzero index matrices (align.py:449-458),
                                                   built ONCE per sequence outside the timed loop
    o, t, z allocation + boundary initialisation   (align.py:357-385)
    cext_align_<mode>(m, g1, g2, o, t, z)
    end cell / score                               (align.py:401-431; global and local here)

usage: ref_baseline.py <batch.npz | -> <worker> <workers> <seconds> <mode>     -> one JSON line on stdout
("-": the path of the batch file arrives as one line on stdin - bench.py starts its workers before it
initialises the GPU and releases them after the timed region; end of input = nothing to do)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import load_ref_cext  # noqa: E402


def nonzero_matrix(p):
    """align.py:449-458: per row the ascending indices of the nonzero entries, -1 padded (intp)."""
    L, A = p.shape
    nz = np.full((L, A), -1, dtype=np.intp)
    mask = p != 0
    order = np.argsort(~mask, axis=1, kind="stable")          # nonzero columns first, ascending
    cnt = mask.sum(axis=1)
    keep = np.arange(A)[None, :] < cnt[:, None]
    nz[keep] = order[keep]
    return nz


def boundaries(mode, g1, g2):
    """align.py:357-385 for the modes the bench times (global / local: penalised edges)."""
    L1, L2 = g1.shape[0], g2.shape[0]
    o = np.zeros((L1 + 1, L2 + 1, 3), dtype=np.float32)
    o[:, 0, :] = -np.inf
    o[0, :, :] = -np.inf
    t = np.zeros((L1 + 1, L2 + 1, 3), dtype=np.uint8)
    z = np.zeros((L1 + 1, L2 + 1), dtype=np.uint8)
    o[0, 0, 0] = 0
    o[0, 0, 1] = g1[0, 0] - g1[0, 1]
    o[1:, 0, 1] = np.arange(L1) * g1[:, 1] + g1[0, 0]
    t[1:, 0, 1] = 32   # UP_EXTEND
    o[0, 0, 2] = g2[0, 0] - g2[0, 1]
    o[0, 1:, 2] = np.arange(L2) * g2[:, 1] + g2[0, 0]
    t[0, 1:, 2] = 128  # LEFT_EXTEND
    return o, t, z


def main():
    path, worker, workers, seconds, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
    if mode not in ("global", "local"):
        raise SystemExit("ref_baseline.py times global / local only")
    ref = load_ref_cext()
    if path == "-":
        path = sys.stdin.readline().strip()
        if not path:
            return
    d = np.load(path)
    arena, row_off, lens, S, pairs = d["arena"], d["row_off"], d["lens"], d["S"], d["pairs"]
    gaps = d["gaps"]
    mine = np.arange(worker, len(pairs), workers)
    profs, nzs = {}, {}
    for s in np.unique(pairs[mine]):
        p = np.ascontiguousarray(arena[row_off[s]:row_off[s] + lens[s]])
        profs[int(s)], nzs[int(s)] = p, nonzero_matrix(p)
    align = getattr(ref, "cext_align_" + mode)
    done, scores, cells = [], [], 0
    t0 = time.perf_counter()
    for k in mine:
        i, j = int(pairs[k, 0]), int(pairs[k, 1])
        p1, p2 = profs[i], profs[j]
        m = np.zeros((p1.shape[0], p2.shape[0]), dtype=np.float32)
        ref.cext_build_scores([p1], [p2], [nzs[i]], [nzs[j]], [S], m)
        g1 = np.empty((p1.shape[0], 2), dtype=np.float32)
        g2 = np.empty((p2.shape[0], 2), dtype=np.float32)
        g1[:], g2[:] = gaps, gaps
        o, t, z = boundaries(mode, g1, g2)
        align(m, g1, g2, o, t, z)
        score = float(o.max()) if mode == "local" else float(o[-1, -1].max())
        done.append(int(k))
        scores.append(score)
        cells += p1.shape[0] * p2.shape[0]
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    print(json.dumps({"worker": worker, "seconds": dt, "cells": int(cells), "pairs": done, "scores": scores}))


if __name__ == "__main__":
    main()
