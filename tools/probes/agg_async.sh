R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_agg_gemm.py -x -q 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -x -q 2>&1 | tail -2
