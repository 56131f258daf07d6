#!/usr/bin/env python3
"""bench.py - GCUPS of the all-pairs profile-profile affine-gap DP hot path (BASELINE.json metric).

A step = one pass of the hot path over one batch: the per-sequence pre-multiply (P . S^T, MFMA) plus the fused
match-score + DP kernel over every pair this rank owns, inputs resident in HBM; with more than one rank the step also
holds the path's one exchange: the RCCL all-gather of the score shards and the reassembly into pair order.

Workloads (--workload overrides the default):
  c2  N=1 default  BASELINE configs[1]: 256 seqs ~400 aa as float profiles (~7 nonzeros / column), all 32 640 pairs
  c4  N>1 default  BASELINE configs[3]: 4 096 seqs ~400 aa float profiles, all 8 386 560 pairs SPLIT over the N ranks
                   (whole columns, balanced by DP cells), all-gather of the score list   -> "scaling": "strong"
  c5  by flag      BASELINE configs[4]: 512 nucleotide seqs ~5 kb (one-hot, 15 x 15 IUPAC matrix), 130 816 pairs
                   split over the N ranks
global mode, gaps -11 / -1, score-only.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAP_OPEN, GAP_EXTEND = -11.0, -1.0
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA (the kernel's match-score MFMAs are f16)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (the fp32-chain variant)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
# MI355X_MICROARCH.md (Wave scheduling; cycle constants, v_fma_f32): a SIMD-32 issues a wave64 VALU instruction over 2
# cycles - 256 CUs x 4 SIMDs x 2.4 GHz / 2.  ONE wave alone only gets an instruction out every ~5.5 cycles (any kind:
# scripts/micro/step_cost.hip, issue_cost.hip), so W resident waves per SIMD can reach min(1, W x 2 / 5.5) of that peak.
NOMINAL_GHZ = 2.4
PEAK_VALU_GINSTR = 256 * 4 * NOMINAL_GHZ / 2.0
WAVE_ISSUE_CYCLES = 5.5


def synth_lengths(rng, n, mu):
    return np.clip(np.rint(rng.normal(mu, 0.1 * mu, n)), 0.5 * mu, 1.5 * mu).astype(int)


def synth_profile(rng, L, A=27):
    """one-hot x5 counts + 6 random extra residues with counts 1-3, normalised as
    ProfileTrack.profile does (praline/container/sequence.py:200-202)."""
    counts = np.zeros((L, A), dtype=np.int64)
    counts[np.arange(L), rng.integers(0, 20, L)] += 5
    for _ in range(6):
        counts[np.arange(L), rng.integers(0, 20, L)] += rng.integers(1, 4, L)
    totals = np.array(counts.sum(axis=1), dtype=np.float32)
    return np.array(counts / totals[:, np.newaxis], dtype=np.float32)


def one_hot(values, A):
    p = np.zeros((len(values), A), dtype=np.float32)
    p[np.arange(len(values)), values] = 1.0
    return p


def make_workload(name):
    """Deterministic synthetic inputs of a BASELINE configuration (identical on every rank)."""
    from praline_amd.matrices import blosum62_matrix, nucleotide_matrix
    if name == "c2":
        rng = np.random.default_rng(2)
        lens = synth_lengths(rng, 256, 400)
        return {"name": "C2", "lens": lens, "profs": [synth_profile(rng, int(L)) for L in lens], "S": blosum62_matrix(),
                "desc": "256 seqs ~400 aa float profiles", "matrix": "BLOSUM62"}
    if name == "c4":
        rng = np.random.default_rng(4)
        lens = synth_lengths(rng, 4096, 400)
        return {"name": "C4", "lens": lens, "profs": [synth_profile(rng, int(L)) for L in lens], "S": blosum62_matrix(),
                "desc": "4096 seqs ~400 aa float profiles", "matrix": "BLOSUM62"}
    if name == "c5":
        rng = np.random.default_rng(5)
        lens = synth_lengths(rng, 512, 5000)
        return {"name": "C5", "lens": lens, "profs": [one_hot(rng.integers(0, 4, int(L)), 15) for L in lens],
                "S": nucleotide_matrix(), "desc": "512 nucleotide seqs ~5 kb one-hot", "matrix": "IUPAC nucleotide 15x15"}
    raise SystemExit("unknown workload %r" % name)


def csrc_digest():
    """sha256 over the kernel / host sources of libpraline_dp.so: profiles/*_latest.json are stamped with it, and a
    figure taken from another build is reported as null instead of a stale number (there is no .git on the GPU box)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "praline_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")) or fn == "Makefile":
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def norm_kernel(name):
    return (name or "").replace("void ", "").replace(" ", "").split("(")[0]


def stamped_counters(kernel, workload):
    """Counter-derived figures of profiles/counters_latest.json (written by scripts/profile_bench.sh): only if they
    were collected on THIS build (csrc digest), THIS kernel instance and THIS workload; None otherwise."""
    path = os.path.join(ROOT, "profiles", "counters_latest.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("csrc_sha256") != csrc_digest() or norm_kernel(d.get("kernel")) != norm_kernel(kernel) or \
            d.get("workload") != workload:
        return None
    return d


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
    ap.add_argument("--steps", type=int, default=50)     # (a step is ~2 ms: 50 + 10 steps keep the default run in seconds while the
    ap.add_argument("--warmup", type=int, default=10)    #  timed region is no longer dominated by the first launches after idle)
    ap.add_argument("--mode", default="global")
    ap.add_argument("--workload", choices=["c2", "c4", "c5"], default=None,
                    help="default: c2 on one GPU (the metric's configuration), c4 split over the ranks on more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the side measurements (variants, e2e, sustained)")
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
    wl_name = args.workload or ("c2" if world == 1 else "c4")

    ref_baseline = None
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and wl_name == "c2"
    if want_cpu:
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
    wl = make_workload(wl_name)
    lens, profs, S = wl["lens"], wl["profs"], wl["S"]
    A = int(S.shape[0])
    n_seqs = len(lens)
    ii, jj = np.triu_indices(n_seqs, k=1)          # the reference's pair order (tree.py:105-129)
    pairs = np.stack([ii, jj], axis=1).astype(np.int32)
    cells = lens[pairs[:, 0]].astype(np.int64) * lens[pairs[:, 1]].astype(np.int64)
    total_cells = int(cells.sum())

    # this rank's shard of the pair list: whole columns (pairs sharing sequence two), balanced by cells
    from praline_amd.allpairs import shard_columns, gather_maps
    shards = shard_columns(lens, pairs, world)
    my_idx = shards[rank]
    my_pairs = pairs[my_idx]
    src, dst, slice_len = gather_maps(shards)

    t_a = time.perf_counter()
    arena = native.Arena(profs, S)
    t_b = time.perf_counter()
    plan = native.Plan(arena, my_pairs)
    t_c = time.perf_counter()
    rank_arena_ms, rank_plan_ms = (t_b - t_a) * 1e3, (t_c - t_b) * 1e3   # host side of the stage (outside the timed steps)
    # two score buffers: step k + 1 computes into the other one while step k's slice is still being gathered / copied out
    d_slices = [torch.zeros(slice_len, dtype=torch.float32, device="cuda") for _ in range(2)]
    gathered = [None, None]
    d_slice = d_slices[0]
    d_all = torch.zeros(slice_len * world, dtype=torch.float32, device="cuda") if dist is not None else None
    # gathered shard slot -> position in the reference's row-major pair order
    if dist is not None:
        d_src = torch.as_tensor(src, device="cuda")
        d_dst = torch.as_tensor(dst, device="cuda")
        d_ordered = [torch.zeros(len(pairs), dtype=torch.float32, device="cuda") for _ in range(2)]
    lib_stream = torch.cuda.ExternalStream(native.stream_handle())
    # the metric's wall time runs from the submission to the scores in HOST memory - every step ends with an
    # asynchronous copy of the stage's score list (this rank's slice on one GPU, the gathered and re-ordered list on
    # several) into page-locked host memory on a second stream, under the next step's kernel (C2: 130 KB)
    copy_stream = torch.cuda.Stream()
    n_out = len(pairs) if dist is not None else slice_len
    h_scores = [torch.empty(n_out, dtype=torch.float32, pin_memory=True) for _ in range(2)]

    def step(k):
        buf = d_slices[k & 1]
        if gathered[k & 1] is not None:
            lib_stream.wait_event(gathered[k & 1])   # the gather / copy that last read this buffer has finished
        arena.premultiply()
        plan.run(args.mode, GAP_OPEN, GAP_EXTEND, d_scores=buf.data_ptr())
        if dist is not None:
            # the exchange step: all ranks obtain every score slice (RCCL all-gather over xGMI)
            cur = torch.cuda.current_stream()
            cur.wait_stream(lib_stream)
            if gathered[k & 1] is not None:
                cur.wait_event(gathered[k & 1])      # (the ordered list of step k - 2 has left for the host)
            dist.all_gather_into_tensor(d_all, buf)
            d_ordered[k & 1][d_dst] = d_all[d_src]   # back into the reference's pair order (tree.py:142-145)
            copy_stream.wait_stream(cur)
            src_t = d_ordered[k & 1]
        else:
            copy_stream.wait_stream(lib_stream)
            src_t = buf
        with torch.cuda.stream(copy_stream):
            h_scores[k & 1].copy_(src_t, non_blocking=True)
            gathered[k & 1] = torch.cuda.Event()
            gathered[k & 1].record(copy_stream)

    def fence():
        if dist is not None:
            dist.barrier()
        native.synchronize()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if dist is not None:
        # untimed sanity check of the exchange: this rank's slice sits at its pairs' positions of the gathered list
        kl = (args.warmup + args.steps - 1) & 1
        last = d_slices[kl]
        if not (torch.equal(d_ordered[kl][torch.as_tensor(my_idx, device="cuda")], last[:len(my_idx)])
                and bool(torch.isfinite(d_ordered[kl]).all())
                and torch.equal(h_scores[kl], d_ordered[kl].cpu())):
            raise SystemExit("bench.py: the gathered score list does not hold this rank's slice at its pairs' positions")

    if dist is None:
        # untimed: the host copy of the last step holds what the kernel wrote
        kl = (args.warmup + args.steps - 1) & 1
        if not (torch.equal(h_scores[kl], d_slices[kl].cpu()) and bool(torch.isfinite(h_scores[kl][:len(my_idx)]).all())):
            raise SystemExit("bench.py: the host copy of the scores differs from the device buffer")
    # separate untimed pass: average DP-kernel duration over a few launches (HIP events on the library's stream)
    kms = []
    for _ in range(5):
        plan.run(args.mode, GAP_OPEN, GAP_EXTEND, d_scores=d_slice.data_ptr())
        kms.append(plan.kernel_ms())
    kernel_ms_avg = float(np.mean(kms))
    kernel_name = plan.kernel_name()
    # per rank: pairs, DP cells, plan creation, DP kernel, exchange (all-gather + reorder, CUDA events around 5 untimed
    # repeats) - the driver computes scaling from `value`; this makes a curve readable (who waits for whom)
    gather_ms = None
    if dist is not None:
        cur = torch.cuda.current_stream()
        native.synchronize(); torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(5):
            dist.all_gather_into_tensor(d_all, d_slice)
            d_ordered[0][d_dst] = d_all[d_src]
        e1.record(cur)
        torch.cuda.synchronize()
        gather_ms = e0.elapsed_time(e1) / 5
    my_cells_rank = int(cells[my_idx].sum())
    mine = {"rank": rank, "pairs": int(len(my_idx)), "cells": my_cells_rank, "arena_ms": rank_arena_ms, "plan_ms": rank_plan_ms,
            "kernel_ms": kernel_ms_avg, "gather_ms": gather_ms, "kernel": kernel_name}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    ms_per_step = elapsed / args.steps * 1e3
    gcups = total_cells / (elapsed / args.steps) / 1e9

    # ---- roofline of the dominant kernel on this rank's slice ----
    my_cells = int(cells[my_idx].sum())
    lsum = int((lens[my_pairs[:, 0]] + lens[my_pairs[:, 1]]).sum())
    alg_bytes = 4.0 * A * lsum + 4.0 * len(my_pairs)      # 4A(L1+L2) + 4 per pair
    alg_flops = 2.0 * A * my_cells                        # 2A flop / cell (MFMA step)
    ksec = kernel_ms_avg * 1e-3
    workload_tag = "%s/%d" % (wl["name"], world)
    ctr = stamped_counters(kernel_name, workload_tag)
    info = arena.info()
    valu_per_step = ctr.get("valu_per_step") if ctr else None
    res = plan.kernel_resources()
    clock_ghz = ctr.get("clock_ghz") if ctr else None   # GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the stamped profile
    roofline = {
        # north_star asks for the HBM roofline; the kernel is NOT HBM-bound:
        # its limiter is VALU issue of the recurrence, see "valu" below.
        "bound": "hbm", "achieved": alg_bytes / ksec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
        "frac": alg_bytes / ksec / 1e9 / PEAK_HBM_GBS,
        # L2-miss bytes per launch from the PMC passes of scripts/profile_bench.sh - null unless they were collected
        # on this very build, kernel instance and workload (profiles/counters_latest.json carries the stamps)
        "traffic": ctr.get("hbm_bytes_per_launch") if ctr else None,
        "kernel": kernel_name, "kernel_ms": kernel_ms_avg, "kernel_gcups": my_cells / ksec / 1e9,
        "bytes_per_cell": alg_bytes / my_cells, "algorithmic_bytes_per_launch": alg_bytes,
        "counters_stamp": ({"csrc_sha256": ctr["csrc_sha256"], "kernel": ctr["kernel"], "workload": ctr["workload"],
                            "source": ctr.get("source")} if ctr else None),
        "mfma": {"achieved": alg_flops / ksec / 1e12, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": alg_flops / ksec / 1e12 / PEAK_F16_MFMA_TFLOPS,
                 "dtype": {1: "f16, exact single term", 2: "f16 hi/lo split (3 terms K-packed into 4 MFMAs per step)",
                           3: "f16 hi/lo split (3 terms, 6 MFMAs per step)"}.get(info["f16_terms"], "f16"),
                 "frac_of_fp32_mfma_peak": alg_flops / ksec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                 "f16_terms": info["f16_terms"], "f16_ranges": info["f16_ranges"]},
        # the vector ALU: valu_per_step is the PMC-measured dynamic count (SQ_INSTS_VALU / wave steps) of the stamped
        # profile; peak = 1024 SIMDs x one wave64 instruction per 2 cycles at the nominal clock (frac_at_measured_clock:
        # at the clock the profiled launch ran at).  What bounds the launch is the issue rate of its resident waves:
        # issue_roof_frac = the share of that peak W waves per SIMD can reach at one instruction per ~5.5 cycles each.
        "valu": ({"achieved": plan.steps * valu_per_step / ksec / 1e9, "peak": PEAK_VALU_GINSTR,
                  "unit": "G wave-instr/s", "frac": plan.steps * valu_per_step / ksec / 1e9 / PEAK_VALU_GINSTR,
                  "frac_at_measured_clock": (plan.steps * valu_per_step / ksec / 1e9 / (PEAK_VALU_GINSTR * clock_ghz / NOMINAL_GHZ)
                                             if clock_ghz else None),
                  "clock_ghz": clock_ghz, "instructions_per_step": ctr.get("insts_per_step"),
                  "valu_per_step": valu_per_step, "steps": plan.steps, "tasks": plan.tasks}
                 if valu_per_step else {"valu_per_step": None, "steps": plan.steps, "tasks": plan.tasks}),
        "vgprs": res["vgprs"] or None, "lds_bytes_per_workgroup": res["lds_bytes"] or None,
        "waves_per_simd": res["waves_per_simd"] or None,
        "issue_roof_frac": (min(1.0, res["waves_per_simd"] * 2.0 / WAVE_ISSUE_CYCLES) if res["waves_per_simd"] else None),
    }
    out = {
        "metric": "GCUPS (DP cell updates/s) all-pairs profile-profile affine align",
        "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        # one GPU runs the metric's configuration (C2); more GPUs split a FIXED list (C4 / C5) -> strong scaling
        "scaling": "strong" if world > 1 else "weak", "workload": wl_name, "vs_baseline": None,
        # arithmetic type of the DP; float-profile match scores come from f16 hi/lo MFMAs with fp32 accumulation
        # (roofline.mfma.dtype), the fp32-chain figure is variants.f32_chain_gcups
        "dtype": "f32", "data": "synthetic",
        # a step = pre-multiply + DP kernel over all pairs (inputs resident in HBM) + the score list copied to page-locked
        # host memory (asynchronously, under the next step); kernel-only: roofline.kernel_ms
        "scores_to_host": True,
        "config": {"workload": "%s: %s, all %d pairs%s, %s, %s, gaps -11/-1, score-only" % (
                       wl["name"], wl["desc"], len(pairs), " split over %d ranks" % world if world > 1 else "",
                       args.mode, wl["matrix"]),
                   "n_seqs": n_seqs, "pairs": int(len(pairs)), "cells": total_cells,
                   "pairs_per_gpu": int(len(my_pairs)), "parallelism": "pairs sharded x%d" % world},
        "roofline": roofline,
        "per_rank": per_rank,
        "cell_imbalance": (max(r["cells"] for r in per_rank) / (sum(r["cells"] for r in per_rank) / len(per_rank))),
    }

    side = rank == 0 and world == 1 and not args.no_variants
    if side:
        def timed(fn, reps=5, sync=fence):
            fn(); sync()
            t_a = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return (time.perf_counter() - t_a) / reps

        # ---- sustained: the same step for >= 1 s (clocks settle; `value` covers only steps x ms_per_step) ----
        n_sus = max(args.steps, int(1.2 / max(elapsed / args.steps, 1e-6)) + 1)
        fence()
        t_a = time.perf_counter()
        for k in range(n_sus):
            step(k)
        fence()
        dt_sus = time.perf_counter() - t_a
        out["sustained"] = {"gcups": total_cells * n_sus / dt_sus / 1e9, "seconds": dt_sus, "steps": n_sus}

        # ---- e2e (SURVEY 8(d) wall time): host profiles -> arena (H2D, pack, pre-multiply) -> plan (host scheduling +
        # upload) -> kernel -> scores back in host memory; PCIe-inclusive, never `value` ----
        reps = []
        for _ in range(6):      # (the first ones still warm the staging buffers and the host caches: arena 0.9 -> 0.56 ms)
            t_a = time.perf_counter()
            # (the plan's host scheduling needs the lengths only: it runs on a second host thread beside the arena's
            # concatenation, upload and packing - native.prepare_schedule_async, as PairwiseBatch.scores_for_pairs does)
            prep = native.prepare_schedule_async([len(p) for p in profs], my_pairs)
            ar2 = native.Arena(profs, S)
            t_b = time.perf_counter()
            pl2 = native.Plan(ar2, my_pairs, prepared=prep)
            t_c = time.perf_counter()
            pl2.run(args.mode, GAP_OPEN, GAP_EXTEND)
            sc2 = pl2.scores()
            t_d = time.perf_counter()
            pl2.close(); ar2.close()
            reps.append((t_b - t_a, t_c - t_b, t_d - t_c))
        best = min(reps, key=sum)
        out["e2e"] = {"gcups": total_cells / sum(best) / 1e9, "ms": sum(best) * 1e3, "arena_ms": best[0] * 1e3,
                      "plan_ms": best[1] * 1e3, "run_and_scores_d2h_ms": best[2] * 1e3,
                      "note": "host float32 profiles in, host scores out (best of 6); arena_ms includes the overlapped host scheduling of the plan"}
        assert np.isfinite(sc2).all()

    # ---- side measurements, driver-timed like `value` (inputs and results resident in HBM); N=1 only ----
    if side and wl_name == "c2":
        variants = {}
        plan_p = native.Plan(arena, my_pairs, want_paths=True)
        dt_paths = timed(lambda: plan_p.run(args.mode, GAP_OPEN, GAP_EXTEND))
        variants["float_profiles_with_paths_kernel"] = plan_p.kernel_name()
        # (local alignments of float profiles keep the strip kernels' chain mode: the pipeline two-pass serves global runs)
        dt_paths_l = timed(lambda: plan_p.run("local", GAP_OPEN, GAP_EXTEND), reps=3)
        plan_p.close()
        variants["float_profiles_with_paths_gcups"] = total_cells / dt_paths / 1e9
        variants["float_profiles_with_paths_local_gcups"] = total_cells / dt_paths_l / 1e9
        # fp32 MFMA chain instead of the f16 split (PRALINE_MATCH_F32)
        native.set_match_mode("f32")
        plan_f = native.Plan(arena, my_pairs)
        dt_f32 = timed(lambda: (arena.premultiply(), plan_f.run(args.mode, GAP_OPEN, GAP_EXTEND)))
        variants["f32_chain_gcups"] = total_cells / dt_f32 / 1e9
        variants["f32_chain_kernel"] = plan_f.kernel_name()
        plan_f.close()
        # the reference's own summation order on the VALU (PRALINE_MATCH_REFERENCE): identical alignments for any profiles
        native.set_match_mode("ref")
        sub = my_pairs          # (the whole workload: k_match_tile + the dense-tile instances)
        sub_cells = int((lens[sub[:, 0]].astype(np.int64) * lens[sub[:, 1]]).sum())
        plan_r = native.Plan(arena, sub)
        dt_ref = timed(lambda: plan_r.run(args.mode, GAP_OPEN, GAP_EXTEND), reps=3)
        variants["reference_order_kernel"] = plan_r.kernel_name()
        plan_r.close()
        plan_r = native.Plan(arena, sub, want_paths=True)
        dt_ref_p = timed(lambda: plan_r.run(args.mode, GAP_OPEN, GAP_EXTEND), reps=3)
        plan_r.close()
        native.set_match_mode(None)
        variants["reference_order_gcups"] = sub_cells / dt_ref / 1e9
        variants["reference_order_with_paths_gcups"] = sub_cells / dt_ref_p / 1e9
        variants["reference_order_sample"] = "%d of %d pairs" % (len(sub), len(my_pairs))
        # plans whose fill reads dense match-score tiles: per-position gap scores (GapScoreModel arrays, the
        # fp32 MFMA chain writes the tiles), and an alphabet of 40 active symbols (no packed operands: the reference-order
        # one-cell-per-thread kernels write them)
        rng_g = np.random.default_rng(5)
        arena_g = native.Arena(profs, S)
        arena_g.set_gap_scores([np.stack([-rng_g.uniform(8.0, 14.0, int(L)), -rng_g.uniform(0.5, 2.0, int(L))], axis=1).astype(np.float32) for L in lens])
        plan_g = native.Plan(arena_g, my_pairs)
        variants["per_position_gaps_gcups"] = total_cells / timed(lambda: plan_g.run_gaps(args.mode), reps=3) / 1e9
        variants["per_position_gaps_kernel"] = plan_g.kernel_name()
        plan_g.close()
        plan_g = native.Plan(arena_g, my_pairs, want_paths=True)
        variants["per_position_gaps_with_paths_gcups"] = total_cells / timed(lambda: plan_g.run_gaps(args.mode), reps=3) / 1e9
        plan_g.close()
        arena_g.close()
        A_w = 40
        S_w = rng_g.normal(0, 3, (A_w, A_w)).astype(np.float32)
        profs_w = []
        for L in lens:
            c = np.zeros((int(L), A_w), dtype=np.float32)
            for _ in range(4):
                c[np.arange(int(L)), rng_g.integers(0, A_w, int(L))] += rng_g.integers(1, 4, int(L))
            profs_w.append((c / c.sum(axis=1, keepdims=True)).astype(np.float32))
        arena_w = native.Arena(profs_w, S_w)
        plan_w = native.Plan(arena_w, my_pairs)
        variants["wide_alphabet_gcups"] = total_cells / timed(lambda: plan_w.run(args.mode, GAP_OPEN, GAP_EXTEND), reps=2) / 1e9
        variants["wide_alphabet_kernel"] = plan_w.kernel_name()
        plan_w.close()
        plan_w = native.Plan(arena_w, my_pairs, want_paths=True)
        variants["wide_alphabet_with_paths_gcups"] = total_cells / timed(lambda: plan_w.run(args.mode, GAP_OPEN, GAP_EXTEND), reps=2) / 1e9
        plan_w.close()
        arena_w.close()
        rng1 = np.random.default_rng(2)
        profs_1h = [one_hot(rng1.integers(0, 20, int(L)), A) for L in lens]
        arena_1h = native.Arena(profs_1h, S)
        plan_1h = native.Plan(arena_1h, my_pairs)
        dt_1h = timed(lambda: (arena_1h.premultiply(), plan_1h.run(args.mode, GAP_OPEN, GAP_EXTEND)))
        variants["onehot_score_only_gcups"] = total_cells / dt_1h / 1e9
        variants["onehot_f16_terms"] = arena_1h.info()["f16_terms"]
        plan_1h.close()
        plan_1hp = native.Plan(arena_1h, my_pairs, want_paths=True)
        dt_1hp = timed(lambda: plan_1hp.run(args.mode, GAP_OPEN, GAP_EXTEND))
        variants["onehot_with_paths_kernel"] = plan_1hp.kernel_name()
        variants["onehot_with_paths_local_gcups"] = total_cells / timed(lambda: plan_1hp.run("local", GAP_OPEN, GAP_EXTEND), reps=3) / 1e9
        plan_1hp.close()
        arena_1h.close()
        variants["onehot_with_paths_gcups"] = total_cells / dt_1hp / 1e9

        # C3 (1024 seqs ~250 aa one-hot, ALL 1 047 552 ordered pairs) with paths: fill + end cells + device traceback
        from praline_amd.allpairs import enumerate_pairs
        rng3 = np.random.default_rng(3)
        l3 = synth_lengths(rng3, 1024, 250)
        a3 = native.Arena([one_hot(rng3.integers(0, 20, int(L)), 27) for L in l3], S)
        i3, j3 = np.divmod(np.arange(1024 * 1024, dtype=np.int64), 1024)
        p3 = np.stack([i3[i3 != j3], j3[i3 != j3]], axis=1).astype(np.int32)
        c3 = int((l3[p3[:, 0]].astype(np.int64) * l3[p3[:, 1]]).sum())
        for m3 in ("global", "local", "semiglobal_both"):
            t_b = time.perf_counter()
            pl3 = native.Plan(a3, p3, want_paths=True)
            variants["c3_path_plan_ms"] = min(variants.get("c3_path_plan_ms", 1e9), (time.perf_counter() - t_b) * 1e3)
            variants["c3_with_paths_%s_gcups" % m3] = c3 / timed(lambda: pl3.run(m3, GAP_OPEN, GAP_EXTEND), reps=3) / 1e9
            pl3.close()
        a3.close()
        # C4: one rank's column shard of the 8-rank split (1.05 M pairs, 1.7e11 cells), float profiles
        w4 = make_workload("c4")
        p4 = enumerate_pairs(4096)
        s4 = p4[shard_columns(w4["lens"], p4, 8)[3]]
        c4 = int((w4["lens"][s4[:, 0]].astype(np.int64) * w4["lens"][s4[:, 1]]).sum())
        a4 = native.Arena(w4["profs"], w4["S"])
        pl4 = native.Plan(a4, s4)
        pl4.close()
        # what one rank of the 8-way split pays host-to-host for its stage: arena (4096 profiles, 177 MB over PCIe, packing,
        # pre-multiply) + plan (host scheduling on the scheduler's thread pool + uploads) + run + its score slice in host memory
        t_a = time.perf_counter()
        a4b = native.Arena(w4["profs"], w4["S"])
        t_b = time.perf_counter()
        pl4 = native.Plan(a4b, s4)
        t_c = time.perf_counter()
        pl4.run("global", GAP_OPEN, GAP_EXTEND)
        sc4 = pl4.scores()
        t_d = time.perf_counter()
        assert np.isfinite(sc4).all()
        variants["c4_rank_share_e2e_ms"] = (t_d - t_a) * 1e3
        variants["c4_rank_share_arena_ms"] = (t_b - t_a) * 1e3
        variants["c4_rank_share_plan_ms"] = (t_c - t_b) * 1e3
        variants["c4_rank_share_run_and_scores_d2h_ms"] = (t_d - t_c) * 1e3
        pl4.close(); a4b.close()
        t_b = time.perf_counter()
        pl4 = native.Plan(a4, s4)
        # (the stage's plan above is the first of the process - cold allocations, sleeping scheduler threads; this one is
        # what every later plan of a pipeline costs)
        variants["c4_rank_share_plan_warm_ms"] = (time.perf_counter() - t_b) * 1e3
        variants["c4_rank_share_float_gcups"] = c4 / timed(lambda: (a4.premultiply(), pl4.run("global", GAP_OPEN, GAP_EXTEND)), reps=3) / 1e9
        variants["c4_rank_share_kernel"] = pl4.kernel_name()
        pl4.close()
        # ... and ALL of C4 on this one GPU (8 386 560 pairs, 1.34e12 cells): the like-for-like base of the N > 1 runs
        t_b = time.perf_counter()
        pl4 = native.Plan(a4, p4)
        variants["c4_all_pairs_plan_ms"] = (time.perf_counter() - t_b) * 1e3
        pl4.close()
        t_b = time.perf_counter()
        pl4 = native.Plan(a4, p4)
        variants["c4_all_pairs_plan_warm_ms"] = (time.perf_counter() - t_b) * 1e3
        c4_all = int((w4["lens"][p4[:, 0]].astype(np.int64) * w4["lens"][p4[:, 1]]).sum())
        variants["c4_all_pairs_one_gpu_gcups"] = c4_all / timed(lambda: (a4.premultiply(), pl4.run("global", GAP_OPEN, GAP_EXTEND)), reps=2) / 1e9
        pl4.close(); a4.close()
        # ... the same shard with ONE-HOT sequences (integer scoring, bit-exact): the match-score lookup kernel
        rng4 = np.random.default_rng(4)
        a4 = native.Arena([one_hot(rng4.integers(0, 20, int(L)), 27) for L in w4["lens"]], w4["S"])
        pl4 = native.Plan(a4, s4)
        for m4 in ("global", "local"):
            variants["c4_rank_share_onehot_%s_gcups" % m4] = c4 / timed(lambda: pl4.run(m4, GAP_OPEN, GAP_EXTEND), reps=3) / 1e9
        variants["c4_rank_share_onehot_kernel"] = pl4.kernel_name()
        pl4.close(); a4.close()
        del w4
        # C5: one column shard of 14 (9 249 pairs, 2.3e11 cells), 5 kb nucleotide sequences
        w5 = make_workload("c5")
        p5 = enumerate_pairs(512)
        s5 = p5[shard_columns(w5["lens"], p5, 14)[5]]
        c5 = int((w5["lens"][s5[:, 0]].astype(np.int64) * w5["lens"][s5[:, 1]]).sum())
        a5 = native.Arena(w5["profs"], w5["S"])
        pl5 = native.Plan(a5, s5)
        variants["c5_shard_gcups"] = c5 / timed(lambda: (a5.premultiply(), pl5.run("global", GAP_OPEN, GAP_EXTEND)), reps=2) / 1e9
        variants["c5_shard_kernel"] = pl5.kernel_name()
        pl5.close()
        # ... and ALL 130 816 pairs of C5 (3.3e12 cells) on this one GPU
        pl5 = native.Plan(a5, p5)
        c5_all = int((w5["lens"][p5[:, 0]].astype(np.int64) * w5["lens"][p5[:, 1]]).sum())
        variants["c5_all_pairs_one_gpu_gcups"] = c5_all / timed(lambda: (a5.premultiply(), pl5.run("global", GAP_OPEN, GAP_EXTEND)), reps=2) / 1e9
        pl5.close(); a5.close()
        del w5
        # the preprofile stage of C3 from Sequences to ProfileTracks (component.build_preprofiles: N (N - 1) alignments with
        # paths per pass, counting on the device paths; host-inclusive wall time)
        from praline_amd import component as comp_, container as ct_
        rng3 = np.random.default_rng(3)
        l3 = synth_lengths(rng3, 1024, 250)
        seqs3 = [ct_.Sequence("s%d" % k, [(ct_.TRACK_ID_INPUT, ct_.PlainTrack(None, ct_.ALPHABET_AA, raw_indices=rng3.integers(0, 20, int(L))))])
                 for k, L in enumerate(l3)]
        for m3, it3 in (("global", 1), ("local", 2)):
            comp_.build_preprofiles(seqs3, ct_.TRACK_ID_INPUT, ct_.blosum62(), mode=m3, waterman_eggert_iterations=it3)
            t_a = time.perf_counter()
            comp_.build_preprofiles(seqs3, ct_.TRACK_ID_INPUT, ct_.blosum62(), mode=m3, waterman_eggert_iterations=it3)
            variants["c3_build_preprofiles_%s_ms" % m3] = (time.perf_counter() - t_a) * 1e3
        del seqs3
        native.pool_trim()
        # batched RawPairwiseAligner (praline_raw_batch_*): requests of 400 x 400 with their own m, g1, g2, inputs resident in
        # HBM; one step = boundary rows + fill (k_rawb_fill) + end cells and paths (k_rawb_trace) of every request
        rng_r = np.random.default_rng(9)
        base_r = []
        for _ in range(8):
            m_r = (rng_r.standard_normal((400, 400)) * 3 - 0.5).astype(np.float32)
            g_r = [np.stack([-rng_r.uniform(5, 12, 400), -rng_r.uniform(0.5, 2, 400)], axis=1).astype(np.float32) for _ in range(2)]
            base_r.append((m_r, g_r[0], g_r[1], None))
        for n_r in (500, 2048):
            rb = native.RawBatch([base_r[k % 8] for k in range(n_r)])
            for mode_r in ("global", "local"):
                dt_r = timed(lambda: rb.run(mode_r), reps=5, sync=lambda: rb.results(paths=False))
                variants["raw_batch_%d_%s_gcups" % (n_r, mode_r)] = rb.cells / dt_r / 1e9
            rb.close()
        variants["raw_batch_kernel"] = "k_rawb_fill<false, false> + k_rawb_trace"
        native.pool_trim()
        variants["note"] = ("driver-timed like `value` (inputs / results in HBM).  with_paths = fill with packed traceback + "
                            "end cells + device traceback; onehot = integer scoring (bit-exact mode); f32_chain = fp32 MFMA "
                            "match scores; reference_order = PRALINE_MATCH_REFERENCE (bit-identical alignments for float "
                            "profiles); c3 = all 1 047 552 ordered pairs with paths (build_preprofiles: sequences in, profile tracks "
                            "out, local = two Waterman-Eggert passes); c4 = shard 3 of 8; c5 = shard 5 of 14; raw_batch_<n> = n RawPairwiseAligner "
                            "requests of 400 x 400 (own m, g1, g2 each) in one submission, scores and paths")
        out["variants"] = variants

    # ---- CPU baseline: the oracle (C restatement of the reference path) on the host cores ----
    if want_cpu:
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
        sample_cells = int(cells[idx].sum())
        plan.run(args.mode, GAP_OPEN, GAP_EXTEND, d_scores=d_slice.data_ptr())
        native.synchronize()
        gpu_all = d_slice.cpu().numpy()
        max_rel = float(np.max(np.abs(gpu_all[idx] - cpu_scores) / np.maximum(1.0, np.abs(cpu_scores))))
        # 1-thread figure on a small sample (SURVEY 8(d) asks for both)
        n1 = int(max(8, min(len(pairs), n_sample * 2.0 / max(threads * dt, 1e-3))))
        idx1 = np.linspace(0, len(pairs) - 1, n1).astype(np.int64)
        t_a = time.perf_counter()
        orc.batch_scores(args.mode, arena_cat, row_off, lens.astype(np.int32), S, pairs[idx1], GAP_OPEN, GAP_EXTEND, threads=1)
        dt1 = time.perf_counter() - t_a
        port = {
            "value": sample_cells / dt / 1e9, "unit": "GCUPS", "cores": threads,
            "kind": "port",
            "sample": "%d of %d pairs (evenly spaced), %.1f s, oracle/praline_oracle.c "
                      "(build_nonzero + build_scores + fill + end cell per pair, OpenMP)" % (
                          n_sample, len(pairs), dt),
            "max_rel_diff_vs_gpu": max_rel,
            "value_1_thread": int(cells[idx1].sum()) / dt1 / 1e9,
        }
        # the REAL reference C extension (oracle/_ref, built from the reference's own cext.c), one child
        # process per core as the reference itself scales; falls back to the port when _ref is absent
        ref = ref_baseline.run(arena_cat, row_off, lens, S, pairs, gpu_all)
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
