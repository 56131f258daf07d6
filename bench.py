#!/usr/bin/env python3
"""bench.py - GCUPS of the all-pairs profile-profile affine-gap DP hot path (BASELINE.json metric).

A step = one pass of the hot path over one batch: the per-sequence pre-multiply (P . S^T, MFMA)
plus the fused match-score + DP kernel over every pair of the batch, inputs resident in HBM.
Workload at N=1: BASELINE configs[1] - 256 sequences ~400 aa as float profiles (~7 nonzeros per
column), all 32 640 pairs, global mode, gaps -11/-1, BLOSUM62.  For N>1 the number
of sequences grows so that the pairs per GPU stay constant (weak scaling); every rank aligns its
cell-balanced slice of the pair list and the score slices are all-gathered over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A = 27
GAP_OPEN, GAP_EXTEND = -11.0, -1.0
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA (the kernel's match-score MFMAs are f16)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (the fp32-chain variant)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
VALU_CYCLES_PER_INST = 4.2     # measured (scripts/micro/valu_dep.hip): one wave64 fp32 VALU op per ~4.2 cycles per SIMD
PAIRS_PER_GPU = 32640          # C2: 256 * 255 / 2


def blosum62():
    from praline_amd.matrices import blosum62_matrix
    return blosum62_matrix()


def synth_lengths(rng, n, mu):
    return np.clip(np.rint(rng.normal(mu, 0.1 * mu, n)), 0.5 * mu, 1.5 * mu).astype(int)


VALU_PER_STEP = 143.0                      # k_dp_split16, 3-term, global: profiles/r01_j_final_pmc_summary.txt
PEAK_VALU_GINSTR = 256 * 4 * 2.4 / 4.0     # 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction


def synth_profile(rng, L):
    """one-hot x5 counts + 6 random extra residues with counts 1-3, normalised as
    ProfileTrack.profile does (praline/container/sequence.py:200-202)."""
    counts = np.zeros((L, A), dtype=np.int64)
    counts[np.arange(L), rng.integers(0, 20, L)] += 5
    for _ in range(6):
        counts[np.arange(L), rng.integers(0, 20, L)] += rng.integers(1, 4, L)
    totals = np.array(counts.sum(axis=1), dtype=np.float32)
    return np.array(counts / totals[:, np.newaxis], dtype=np.float32)


def cpu_threads():
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # the GPU box gives one GPU a share of 16 host cores; do not oversubscribe it
    return int(os.environ.get("BENCH_CPU_THREADS", min(threads, 16)))


class ReferenceCpuBaseline:
    """cpu_baseline kind "reference": oracle/ref_baseline.py children run the reference's own compiled cext
    (oracle/_ref) over evenly spaced pairs, one process per core as the reference itself scales.  The
    children are started BEFORE this process initialises the GPU (no fork/exec afterwards), idle on stdin
    during the timed region and are released by `run`; they use neither the GPU nor /root/reference."""

    def __init__(self, workers, seconds, mode):
        import glob
        import subprocess
        self.workers, self.seconds, self.mode, self.procs = workers, seconds, mode, []
        if mode not in ("global", "local") or not glob.glob(os.path.join(ROOT, "oracle", "_ref", "cext*.so")):
            return
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        self.procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "ref_baseline.py"), "-",
                                        str(w), str(workers), str(seconds), mode],
                                       stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for w in range(workers)]

    def close(self):
        for p in self.procs:
            if p.poll() is None:
                p.stdin.close()
        for p in self.procs:
            p.wait()
        self.procs = []

    def run(self, arena_cat, row_off, lens, S, pairs, gpu_scores):
        if not self.procs:
            return None
        import tempfile
        n = int(min(len(pairs), self.workers * 4096))
        idx = np.linspace(0, len(pairs) - 1, n).astype(np.int64)
        outs = []
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "batch.npz")
            np.savez(path, arena=arena_cat, row_off=row_off, lens=lens, S=S, pairs=pairs[idx],
                     gaps=np.array([GAP_OPEN, GAP_EXTEND], dtype=np.float32))
            for p in self.procs:
                p.stdin.write((path + "\n").encode())
                p.stdin.flush()
            for p in self.procs:
                so, _ = p.communicate()
                if p.returncode == 0 and so.strip():
                    outs.append(json.loads(so.decode().strip().splitlines()[-1]))
        ok = len(outs) == len(self.procs)
        self.procs = []
        if not ok:
            return None
        done = np.concatenate([np.asarray(o["pairs"], dtype=np.int64) for o in outs])
        ref_scores = np.concatenate([np.asarray(o["scores"], dtype=np.float64) for o in outs])
        max_rel = float(np.max(np.abs(gpu_scores[idx[done]] - ref_scores) / np.maximum(1.0, np.abs(ref_scores))))
        return {
            "value": sum(o["cells"] / o["seconds"] for o in outs) / 1e9, "unit": "GCUPS", "cores": self.workers,
            "kind": "reference",
            "sample": "%d of %d pairs (evenly spaced), %d processes x %.1f s, oracle/_ref = the reference's own "
                      "praline/util/cext.c (cext_build_scores + cext_align_%s + numpy boundary init / end cell "
                      "per pair; nonzero index matrices prebuilt per sequence)" % (
                          len(done), len(pairs), self.workers, max(o["seconds"] for o in outs), self.mode),
            "max_rel_diff_vs_gpu": max_rel,
            "value_1_process": float(np.mean([o["cells"] / o["seconds"] for o in outs])) / 1e9,
        }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="global")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the one-hot / with-path side measurements")
    ap.add_argument("--cpu-sample-per-thread", type=int, default=24)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"  # exercise the RCCL exchange with 1 rank
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    ref_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_baseline = ReferenceCpuBaseline(cpu_threads(), args.cpu_seconds, args.mode)

    import torch
    from praline_amd import native

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    native.init(local_rank)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    # ---- synthetic batch (identical on every rank) ----
    n_seqs = int(round(0.5 + math.sqrt(0.25 + 2.0 * world * PAIRS_PER_GPU)))
    rng = np.random.default_rng(2)
    lens = synth_lengths(rng, n_seqs, 400)
    profs = [synth_profile(rng, int(L)) for L in lens]
    S = blosum62()
    pairs = np.array([(i, j) for i in range(n_seqs) for j in range(i + 1, n_seqs)], dtype=np.int32)
    cells = lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]].astype(np.int64)
    total_cells = int(cells.sum())

    # this rank's shard of the pair list: whole columns (pairs sharing sequence two), balanced by cells
    from praline_amd.allpairs import shard_columns, gather_maps
    shards = shard_columns(lens, pairs, world)
    my_idx = shards[rank]
    my_pairs = pairs[my_idx]
    src, dst, slice_len = gather_maps(shards)

    arena = native.Arena(profs, S)
    plan = native.Plan(arena, my_pairs)
    # two score buffers: step k + 1 computes into the other one while step k's slice is still being gathered
    d_slices = [torch.zeros(slice_len, dtype=torch.float32, device="cuda") for _ in range(2)]
    gathered = [None, None]
    d_slice = d_slices[0]
    d_all = torch.zeros(slice_len * world, dtype=torch.float32, device="cuda") if dist is not None else None
    # gathered shard slot -> position in the reference's row-major pair order
    if dist is not None:
        d_src = torch.as_tensor(src, device="cuda")
        d_dst = torch.as_tensor(dst, device="cuda")
        d_ordered = torch.zeros(len(pairs), dtype=torch.float32, device="cuda")
    lib_stream = torch.cuda.ExternalStream(native.stream_handle())

    def step(k):
        buf = d_slices[k & 1]
        if gathered[k & 1] is not None:
            lib_stream.wait_event(gathered[k & 1])   # the gather that last read this buffer has finished
        arena.premultiply()
        plan.run(args.mode, GAP_OPEN, GAP_EXTEND, d_scores=buf.data_ptr())
        if dist is not None:
            # the exchange step: all ranks obtain every score slice (RCCL all-gather over xGMI)
            cur = torch.cuda.current_stream()
            cur.wait_stream(lib_stream)
            dist.all_gather_into_tensor(d_all, buf)
            d_ordered[d_dst] = d_all[d_src]   # back into the reference's pair order (tree.py:142-145)
            gathered[k & 1] = torch.cuda.Event()
            gathered[k & 1].record(cur)

    def fence():
        if dist is not None:
            dist.barrier()
        native.synchronize()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms.append(plan.kernel_ms())  # HIP events around the last DP launch on its own stream
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if dist is not None:
        # untimed sanity check of the exchange: this rank's slice sits at its pairs' positions of the gathered list
        last = d_slices[(args.warmup + args.steps - 1) & 1]
        if not (torch.equal(d_ordered[torch.as_tensor(my_idx, device="cuda")], last[:len(my_idx)])
                and bool(torch.isfinite(d_ordered).all())):
            raise SystemExit("bench.py: the gathered score list does not hold this rank's slice at its pairs' positions")

    # separate untimed pass: average DP-kernel duration over a few launches via HIP events
    kms = []
    for _ in range(5):
        plan.run(args.mode, GAP_OPEN, GAP_EXTEND, d_scores=d_slice.data_ptr())
        kms.append(plan.kernel_ms())
    kernel_ms_avg = float(np.mean(kms))

    ms_per_step = elapsed / args.steps * 1e3
    gcups = total_cells / (elapsed / args.steps) / 1e9

    # ---- roofline of the dominant kernel (k_dp_split16) on this rank's slice ----
    my_cells = int(cells[my_idx].sum())
    lsum = int((lens[my_pairs[:, 0]] + lens[my_pairs[:, 1]]).sum())
    alg_bytes = 4.0 * A * lsum + 4.0 * len(my_pairs)      # 4A(L1+L2) + 4 per pair
    alg_flops = 2.0 * A * my_cells                        # 2A flop / cell (MFMA step)
    ksec = kernel_ms_avg * 1e-3
    traffic = None
    traffic_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(traffic_file):
        try:
            traffic = json.load(open(traffic_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    info = arena.info()
    roofline = {
        # north_star asks for the HBM roofline; the kernel is NOT HBM-bound:
        # its limiter is VALU issue of the recurrence, see "valu" below.
        "bound": "hbm", "achieved": alg_bytes / ksec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": alg_bytes / ksec / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
        "kernel": "k_dp_split16", "kernel_ms": kernel_ms_avg, "kernel_gcups": my_cells / ksec / 1e9,
        "bytes_per_cell": alg_bytes / my_cells,
        "mfma": {"achieved": alg_flops / ksec / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": alg_flops / ksec / 1e12 / PEAK_F16_MFMA_TFLOPS, "dtype": "f16 hi/lo split (3 terms)",
                 "frac_of_fp32_mfma_peak": alg_flops / ksec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                 "f16_terms": info["f16_terms"], "f16_ranges": info["f16_ranges"]},
        # what actually bounds the recurrence: VALU issue.  VALU_PER_STEP is the PMC-measured dynamic count
        # (profiles/r01_j_*: SQ_INSTS_VALU / steps); peak = 1024 SIMDs x one wave64 instruction per 4 cycles.
        "valu": {"achieved": plan.steps * VALU_PER_STEP / ksec / 1e9, "peak": PEAK_VALU_GINSTR,
                 "unit": "G wave-instr/s", "frac": plan.steps * VALU_PER_STEP / ksec / 1e9 / PEAK_VALU_GINSTR,
                 "valu_per_step": VALU_PER_STEP, "steps": plan.steps, "tasks": plan.tasks},
    }
    out = {
        "metric": "GCUPS (DP cell updates/s) all-pairs profile-profile affine align",
        "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: %d seqs ~400 aa float profiles, all %d pairs, %s, BLOSUM62, "
                               "gaps -11/-1, score-only" % (n_seqs, len(pairs), args.mode),
                   "n_seqs": n_seqs, "pairs": int(len(pairs)), "cells": total_cells,
                   "pairs_per_gpu": int(len(my_pairs)), "parallelism": "pairs sharded x%d" % world},
        "roofline": roofline,
    }

    # ---- side measurements on the same batch shape (one-hot variant, with-path run); N=1 only,
    # not part of `value`.  Same timing rule: inputs and results stay in HBM.
    if rank == 0 and world == 1 and not args.no_variants:
        def timed(fn, reps=5):
            fn(); fence()
            t_a = time.perf_counter()
            for _ in range(reps):
                fn()
            fence()
            return (time.perf_counter() - t_a) / reps

        variants = {}
        dt_paths = None
        plan_p = native.Plan(arena, my_pairs, want_paths=True)
        dt_paths = timed(lambda: plan_p.run(args.mode, GAP_OPEN, GAP_EXTEND))
        plan_p.close()
        variants["float_profiles_with_paths_gcups"] = total_cells / dt_paths / 1e9
        rng1 = np.random.default_rng(2)
        profs_1h = [np.eye(A, dtype=np.float32)[rng1.integers(0, 20, int(L))] for L in lens]
        arena_1h = native.Arena(profs_1h, S)
        plan_1h = native.Plan(arena_1h, my_pairs)
        dt_1h = timed(lambda: (arena_1h.premultiply(), plan_1h.run(args.mode, GAP_OPEN, GAP_EXTEND)))
        variants["onehot_score_only_gcups"] = total_cells / dt_1h / 1e9
        variants["onehot_f16_terms"] = arena_1h.info()["f16_terms"]
        plan_1h.close()
        plan_1hp = native.Plan(arena_1h, my_pairs, want_paths=True)
        dt_1hp = timed(lambda: plan_1hp.run(args.mode, GAP_OPEN, GAP_EXTEND))
        plan_1hp.close()
        arena_1h.close()
        variants["onehot_with_paths_gcups"] = total_cells / dt_1hp / 1e9
        variants["note"] = ("same 32640 pairs; with_paths = fill with packed traceback + end cells + device traceback, "
                            "paths left in HBM; onehot = integer scoring (bit-exact mode)")
        out["variants"] = variants

    # ---- CPU baseline: the oracle (C restatement of the reference path) on the host cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        threads = cpu_threads()
        arena_cat = np.concatenate(profs, axis=0)
        row_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)

        def run_sample(n):
            idx_ = np.linspace(0, len(pairs) - 1, n).astype(np.int64)
            t_a = time.perf_counter()
            sc_ = orc.batch_scores(args.mode, arena_cat, row_off, lens.astype(np.int32), S, pairs[idx_],
                                   GAP_OPEN, GAP_EXTEND, threads=threads)
            return idx_, sc_, time.perf_counter() - t_a

        # pilot to size the sample: about args.cpu_seconds of CPU work (bounded by the whole pair list)
        n_pilot = min(len(pairs), threads * args.cpu_sample_per_thread)
        idx, cpu_scores, dt = run_sample(n_pilot)
        n_sample = int(min(len(pairs), max(n_pilot, n_pilot * args.cpu_seconds / max(dt, 1e-3))))
        if n_sample > n_pilot:
            idx, cpu_scores, dt = run_sample(n_sample)
        tc0, tc1 = 0.0, dt
        sample_cells = int(cells[idx].sum())
        gpu_scores = d_slice.cpu().numpy()[idx] if world == 1 else None
        max_rel = float(np.max(np.abs(gpu_scores - cpu_scores) / np.maximum(1.0, np.abs(cpu_scores))))
        # 1-thread figure on a small sample (SURVEY 8(d) asks for both)
        n1 = int(max(8, min(len(pairs), n_sample * 2.0 / max(threads * (tc1 - tc0), 1e-3))))
        idx1 = np.linspace(0, len(pairs) - 1, n1).astype(np.int64)
        t_a = time.perf_counter()
        orc.batch_scores(args.mode, arena_cat, row_off, lens.astype(np.int32), S, pairs[idx1], GAP_OPEN, GAP_EXTEND, threads=1)
        dt1 = time.perf_counter() - t_a
        port = {
            "value": sample_cells / (tc1 - tc0) / 1e9, "unit": "GCUPS", "cores": threads,
            "kind": "port",
            "sample": "%d of %d pairs (evenly spaced), %.1f s, oracle/praline_oracle.c "
                      "(build_nonzero + build_scores + fill + end cell per pair, OpenMP)" % (
                          n_sample, len(pairs), tc1 - tc0),
            "max_rel_diff_vs_gpu": max_rel,
            "value_1_thread": int(cells[idx1].sum()) / dt1 / 1e9,
        }
        # the REAL reference C extension (oracle/_ref, built from the reference's own cext.c), one child
        # process per core as the reference itself scales; falls back to the port when _ref is absent
        ref = ref_baseline.run(arena_cat, row_off, lens, S, pairs, d_slice.cpu().numpy())
        if ref is not None:
            ref["port"] = port
            out["cpu_baseline"] = ref
        else:
            out["cpu_baseline"] = port
    if ref_baseline is not None:
        ref_baseline.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
