#!/bin/bash
# PMC counters of the scores kernel on one rank's share of C4 (scripts/exp_c4.py), one counter group per pass.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_c4${ONEHOT:+_onehot}
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  GS=-1 rocprofv3 --pmc $set --output-format csv -d $OUT/$name -- python3 scripts/exp_c4.py > $OUT/$name.log 2>&1 || { echo "pmc $set failed"; tail -3 $OUT/$name.log; continue; }
  echo "pass $name done"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dp_split16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s per launch %.5g  (%d samples)" % (k, sum(v) / len(v), len(v)))
PY
