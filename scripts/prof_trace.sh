#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/prof_trace.sh <tag> <script.py> [args...]
# rocprofv3 --kernel-trace --stats of `python3 <script.py> args` -> gpurun_out/prof_<tag>/kernel_stats.csv (+ run.log);
# PMC=1 adds the counter passes of scripts/profile_bench.sh (one group per pass, never combined with a trace domain)
# -> gpurun_out/prof_<tag>/pmc_summary.txt (per-kernel sums over the run's dispatches).
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-100s calls %4s avg %10.1f us  total %8.2f ms" % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $OUT/trace
if [ "$PMC" = "1" ]; then
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python3 "$@" > $OUT/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -3 $OUT/pmc_$name.log; continue; }
  done
  python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        a = agg[row['Kernel_Name'].split('(')[0][:90]][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
with open(out + '/pmc_summary.txt', 'w') as fo:
    for k, d in agg.items():
        if max(v[0] for v in d.values()) < 1e6: continue
        fo.write("== %s\n" % k)
        for c, (v, n) in d.items(): fo.write("  %-28s per launch %.6g  (launches %d)\n" % (c, v / n, n))
        if 'SQ_WAVE_CYCLES' in d:
            wc = d['SQ_WAVE_CYCLES'][0]
            fo.write("  # share of wave cycles: wait_any %.2f wait_inst %.2f active_any %.2f active_valu %.2f\n" % (
                d['SQ_WAIT_ANY'][0] / wc, d['SQ_WAIT_INST_ANY'][0] / wc, d['SQ_ACTIVE_INST_ANY'][0] / wc, d['SQ_ACTIVE_INST_VALU'][0] / wc))
print(open(out + '/pmc_summary.txt').read())
PY
  rm -rf $OUT/pmc_*/
fi
