#!/bin/bash
# usage: scripts/profile_rawb.sh <tag>   (run on the GPU box via gpurun)
# Kernel trace + PMC counters of the batched RawPairwiseAligner (scripts/exp_raw_batch.py --bench-only --sizes 2048x400):
# per-launch averages per kernel -> gpurun_out/prof_<tag>/{kernel_stats.csv,summary.txt}.
set -e
TAG=${1:-rawb}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="scripts/exp_raw_batch.py --bench-only --sizes 2048x400"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || (tail -20 $OUT/trace.log; exit 1)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
grep GCUPS $OUT/trace.log
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -3 $OUT/pmc_$name.log; continue; }
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:48]
        a = agg[k][row['Counter_Name']]; a[0] += float(row['Counter_Value']); a[1] += 1
with open(out + '/summary.txt', 'w') as fo:
    fo.write("# rocprofv3 --pmc, one counter group per pass; per-launch averages; 2048 requests of 400 x 400 (327.68 M cells per launch)\n")
    for k, cs in agg.items():
        if 'k_rawb' not in k: continue
        fo.write("== %s\n" % k)
        for c, (v, n) in cs.items(): fo.write("  %-28s %.6g\n" % (c, v / n))
        if 'SQ_WAVE_CYCLES' in cs:
            wc = cs['SQ_WAVE_CYCLES'][0] / cs['SQ_WAVE_CYCLES'][1]
            g = lambda c: cs[c][0] / cs[c][1] / wc if c in cs else float('nan')
            fo.write("  # share of wave cycles: wait_any %.2f wait_inst %.2f active_any %.2f active_valu %.2f\n" % (g('SQ_WAIT_ANY'), g('SQ_WAIT_INST_ANY'), g('SQ_ACTIVE_INST_ANY'), g('SQ_ACTIVE_INST_VALU')))
print(open(out + '/summary.txt').read())
PY
