# A/B of the weight-operand pre-split of the three-limb NN GEMM (CB_LIMB_PRESPLIT=1 = split once per launch + LDS-DMA), with PMC
# instruction counts of the same runs.  usage (GPU box): bash tools/probes/gemm_presplit_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export CB_LIMB_PRESPLIT=$v
  echo "=== PRESPLIT=$v"
  python $R/tools/bench_gemm.py 2>&1 | grep "NN M" | grep -v addend
  for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_ps${v}_$tag -- python $R/tools/gemm_one.py 10000000 256 256 > /dev/null 2>&1
    f=$(find /tmp/pmc_ps${v}_$tag -name "*counter_collection.csv" | head -1)
    echo "--- $c"; python $R/tools/pmc_summary.py $f k_gemm_nn_l3 2>&1 | head -8
  done
done
