cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in base ra1 ra2; do
  if [ $v = base ]; then unset PRALINE_LIB; else export PRALINE_LIB=$GRAFT_REPO_ROOT/variants/libpraline_dp_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$v -- python3 scripts/exp_raw_batch.py --bench-only --sizes 2048x400 > gpurun_out/abl_$v.log 2>&1
  f=$(find gpurun_out/abl_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && grep "fill\|trace" "$f" | cut -d, -f1,4 
done
