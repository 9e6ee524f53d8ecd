R=$GRAFT_REPO_ROOT
cd $R
CB_AGG_GEMM_TRUNKBWD=1 timeout 900 python -m pytest tests/test_gpu_agg_gemm.py -x -q 2>&1 | tail -2
for e in "CB_AGG_GEMM_TRUNKBWD=0" "CB_AGG_GEMM_TRUNKBWD=1"; do
  echo "--- $e"
  env $e timeout 600 python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('final_loss'))"
done
