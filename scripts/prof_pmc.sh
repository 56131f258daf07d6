#!/bin/bash
# usage (GPU box): scripts/prof_pmc.sh <tag> "<counter set 1>;<counter set 2>;..." <script.py> [args]   -> gpurun_out/pmc_<tag>.txt
TAG=$1; SETS=$2; shift 2
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra ARR <<< "$SETS"
i=0
for set in "${ARR[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || { echo "pmc $set failed"; tail -3 $OUT/p$i.log; }
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        a = agg[row['Kernel_Name'].split('(')[0][:80]][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
for k, d in agg.items():
    if max(v[0] / max(v[1], 1) for v in d.values()) < 1e6: continue
    print("==", k)
    for c, (v, n) in sorted(d.items()): print("  %-36s per launch %.6g (n=%d)" % (c, v / n, n))
PY
rm -rf $OUT/p*/
