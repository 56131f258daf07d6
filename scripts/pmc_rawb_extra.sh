#!/bin/bash
# extra counter passes for k_rawb_fill (instruction fetch, LDS waits); run on the GPU box
OUT=$PWD/gpurun_out/prof_rawb_extra
mkdir -p $OUT; export TMPDIR=/tmp
ARGS="scripts/exp_raw_batch.py --bench-only --sizes 2048x400"
for set in "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "SQC_DCACHE_REQ SQC_DCACHE_MISSES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -2 $OUT/pmc_$name.log | cut -c1-300; continue; }
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:48]
        a = agg[k][row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
for k, cs in agg.items():
    if 'fill' not in k: continue
    print("==", k)
    for c, (v, n) in cs.items(): print("  %-32s %.6g" % (c, v / n))
PY
