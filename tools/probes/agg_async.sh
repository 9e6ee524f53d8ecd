R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_agg_gemm.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
bash tools/probes/ab.sh 2 "CB_TRUNK_X0_BITS=0" "CB_TRUNK_X0_BITS=1"
