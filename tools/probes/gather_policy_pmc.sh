set -x
python -m pytest tests/test_gpu_graph_spmm.py tests/test_gpu_fullsize.py tests/test_gpu_random.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
for g in 0 2; do CB_SPMM_GATHER=$g python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --ref-epochs 0 > gpurun_out/r02_gp${g}_bench.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r02_gp${g}_bench.json'));print('GATHER=$g', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
cd /tmp && export TMPDIR=/tmp
for g in 0 2; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum"; do
    tag=$(echo $c | tr ' ' '_')
    CB_SPMM_GATHER=$g rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gp${g}_$tag -- python $GRAFT_REPO_ROOT/tools/bench_spmm.py --iters 3 > /dev/null 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_gp${g}_$tag -name "*counter_collection.csv" | head -1)
    echo "--- GATHER=$g $c"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f k_spmm 2>&1 | head -20
  done
done
