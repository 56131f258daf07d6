#!/bin/bash
# usage: scripts/profile_bench.sh <tag>   (run on the GPU box via gpurun)
# kernel-trace stats first; PMC counters in their own runs (never combined with trace domains).
set -e
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1 || (tail -20 $OUT/bench_trace.log; exit 1)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -5 $OUT/pmc_$name.log; continue; }
  f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    k = (row.get('Kernel_Name', '')[:40], row.get('Counter_Name'))
    agg[k][0] += float(row.get('Counter_Value', 0)); agg[k][1] += 1
for (kn, cn), (v, n) in sorted(agg.items()):
    if "k_dp_" in kn:
        print("%-42s %-28s per-launch %.4g (n=%d)" % (kn, cn, v / n, n))
PY
done
