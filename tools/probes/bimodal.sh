# the step time of the same binary on one box is bimodal (≈ 197.5 / 200.3 ms): eight runs, per-kernel means of the two largest kernels beside it
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('run $i  %.2f ms  fused avg %.3f ms  plain avg %.3f ms  peak %.1f GB' % (d['ms_per_step'], r['aggregation_plus_gemm_launches']['avg_launch_ms'], r['plain_aggregation_launches']['avg_launch_ms'], d['peak_mem_gb']))"
done
