R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_agg_gemm.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
for i in 1 2; do timeout 300 python tools/bench_agg_gemm.py --iters 5 --parts 0 2>&1 | tail -1; CB_AGG_GEMM_DBG=8 timeout 300 python tools/bench_agg_gemm.py --iters 5 --parts 0 2>&1 | tail -1; done
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('final_loss'))"
