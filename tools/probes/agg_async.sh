R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
for v in 0 1; do
  ( cd /tmp && CB_TRUNK_PREMASKED=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_pm$v -o pm$v -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > /dev/null 2>&1 )
  DB=$(find /tmp/prof_pm$v -name "*.db" | head -1)
  python tools/prof_summary.py $DB /tmp/pm$v.md "premasked=$v" > /dev/null 2>&1
  echo "=== premasked=$v"; grep -E "k_agg_gemm2|input_bwd_multi|k_gemm_nn_l3<2, 2, false, 4, 1, 2|trunk_bwd" /tmp/pm$v.md | cut -c1-140
done
