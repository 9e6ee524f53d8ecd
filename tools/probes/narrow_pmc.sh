# HBM-side traffic of the narrow-width aggregation (d = 16 / 40 / 47 / 64) on the ogbn-products shape: is it bound by edges or by sectors?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in ${WIDTHS:-16 40 47 48 64}; do
  echo "=== d=$d"
  python $R/tools/bench_spmm.py --name S-products --n 2449029 --d $d --iters 5 2>&1 | tail -1
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_nw_${d}_$tag -- python $R/tools/bench_spmm.py --name S-products --n 2449029 --d $d --iters 2 > /dev/null 2>&1
    f=$(find /tmp/pmc_nw_${d}_$tag -name "*counter_collection.csv" | head -1)
    python $R/tools/pmc_summary.py $f k_spmm_rows k_spmm_small 2>&1 | grep -v "^$" | head -6
  done
done
