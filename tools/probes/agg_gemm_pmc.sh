# PMC counters of the aggregation + GEMM kernel (k_agg_gemm2) beside its two parts (k_spmm_rows, k_gemm_nn_l3), separate passes.
# usage (GPU box): bash tools/probes/agg_gemm_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_ag_$tag -- python $R/tools/bench_agg_gemm.py --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_ag_$tag -name "*counter_collection.csv" | head -1)
  echo "--- $c"; python $R/tools/pmc_summary.py $f k_agg_gemm2 k_spmm_rows k_gemm_nn_l3 k_spmm_hub_chunks 2>&1
done
