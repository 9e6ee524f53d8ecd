R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_agg_gemm.py tests/test_gpu_model.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for e in "CB_TRUNK_FUSE_OUT_BWD=0" "CB_TRUNK_FUSE_OUT_BWD=1"; do
  echo "--- $e"
  env $e timeout 600 python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d.get('final_loss'))"
done
