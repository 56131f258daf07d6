#!/bin/bash
# usage: scripts/profile_bench.sh <tag>   (run on the GPU box via gpurun)
# 1. rocprofv3 --kernel-trace --stats of the bench command; 2. PMC counters in their own passes (never combined with
# trace domains); 3. writes gpurun_out/prof_<tag>/{kernel_stats.csv,summary.txt,counters.json}.  counters.json is
# STAMPED with the sha256 of praline_amd/csrc, the kernel instance and the workload; copy it to
# profiles/counters_latest.json (and the rest to profiles/<round>_<tag>_*) - bench.py reports roofline.traffic and
# roofline.valu only when the stamps match the library it runs.
set -e
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-variants > $OUT/bench_trace.log 2>&1 || (tail -20 $OUT/bench_trace.log; exit 1)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -6 $OUT/kernel_stats.csv | cut -c1-200
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $OUT/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -3 $OUT/pmc_$name.log; continue; }
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, json, collections, os
sys.path.insert(0, os.getcwd())
import bench
out = sys.argv[1]
line = [l for l in open(out + '/bench_trace.log') if l.startswith('{"metric"')][-1]
b = json.loads(line)
kernel = b["roofline"]["kernel"]
steps = b["roofline"]["valu"]["steps"]
key = bench.norm_kernel(kernel)
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if bench.norm_kernel(row['Kernel_Name']) == key:
            agg[row['Counter_Name']][0] += float(row['Counter_Value']); agg[row['Counter_Name']][1] += 1
    for k, (v, n) in agg.items(): rows[k] = v / n
trace_avg_ns = None
for row in csv.DictReader(open(out + '/kernel_stats.csv')):
    if bench.norm_kernel(row['Name']) == key:
        trace_avg_ns = float(row['AverageNs'])
with open(out + '/summary.txt', 'w') as fo:
    fo.write("# rocprofv3 --pmc, one counter group per pass, bench.py --steps 3 --warmup 1; per-launch averages of %s\n" % kernel)
    fo.write("# csrc sha256 %s, workload %s, kernel-trace average %.4f ms, bench HIP-event average %.4f ms\n" % (
        bench.csrc_digest(), b["config"]["workload"].split(":")[0], (trace_avg_ns or 0) / 1e6, b["roofline"]["kernel_ms"]))
    for k, v in rows.items(): fo.write("%-32s %.6g\n" % (k, v))
    if 'SQ_WAVE_CYCLES' in rows:
        wc = rows['SQ_WAVE_CYCLES']
        fo.write("# share of wave cycles: wait_any %.2f  wait_inst %.2f  active_any %.2f  active_valu %.2f\n" % (
            rows['SQ_WAIT_ANY'] / wc, rows['SQ_WAIT_INST_ANY'] / wc, rows['SQ_ACTIVE_INST_ANY'] / wc, rows['SQ_ACTIVE_INST_VALU'] / wc))
    if 'SQ_INSTS_VALU' in rows:
        fo.write("# VALU instructions per wavefront step (1024 cells): %.1f  (steps per launch %d)\n" % (rows['SQ_INSTS_VALU'] / steps, steps))
res = {"kernel": kernel, "workload": "%s/%d" % (b["config"]["workload"].split(":")[0], b["n_gpus"]),
       "csrc_sha256": bench.csrc_digest(), "source": out, "kernel_trace_avg_ms": (trace_avg_ns or 0) / 1e6,
       "bench_kernel_ms": b["roofline"]["kernel_ms"], "steps_per_launch": steps,
       "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
if 'SQ_INSTS_VALU' in rows:
    res["valu_per_step"] = rows['SQ_INSTS_VALU'] / steps
    # every instruction a wave issues per step (VALU incl. MFMA, SALU, LDS, VMEM): what the per-wave issue rate prices
    res["insts_per_step"] = sum(rows.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR')) / steps
if 'GRBM_GUI_ACTIVE' in rows and trace_avg_ns:
    res["clock_ghz"] = rows['GRBM_GUI_ACTIVE'] / 8.0 / trace_avg_ns
if 'FETCH_SIZE' in rows and 'WRITE_SIZE' in rows:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
    # of the bytes of wide (16 B / lane) coalesced reads -> doubled; WRITE_SIZE is exact for wide stores.
    fetch, write = rows['FETCH_SIZE'] * 1024, rows['WRITE_SIZE'] * 1024
    res.update({"fetch_size_bytes_raw": fetch, "write_size_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write,
                "traffic_over_algorithmic": (2 * fetch + write) / b["roofline"]["algorithmic_bytes_per_launch"],
                "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); L2-miss "
                        "traffic, includes Infinity-Cache hits"})
json.dump(res, open(out + '/counters.json', 'w'), indent=1)
print(open(out + '/summary.txt').read())
print(json.dumps(res))
PY
