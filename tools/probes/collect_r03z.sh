#!/bin/bash
# Round-3 final evidence: full GPU suite, default bench line (cpu_baseline + PMC traffic), rocprofv3 kernel stats + step trace of the same tree,
# fused-kernel decomposition under the flag hand-over, other shapes.
mkdir -p gpurun_out/r03z
O=gpurun_out/r03z
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r03z -o r03z -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > /dev/null 2>&1 )
DB=$(find /tmp/prof_r03z -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 (S-pl10M, N=1, final round-3 tree: flag hand-over in the aggregation + GEMM kernels)" > /dev/null 2>&1
python tools/step_trace.py $DB > $O/step_trace.txt 2>&1
python tools/bench_agg_gemm.py > $O/agg_gemm_parts.txt 2>&1
for v in "SYNC=barrier U=8" "SYNC=flags U=8" "SYNC=flags U=12" "SYNC=flags U=16"; do
  set -- $v
  echo "hand-over $1 $2" >> $O/agg_gemm_variants.txt
  for d in 0 8 16 1 2 3; do echo "  dbg=$d" >> $O/agg_gemm_variants.txt; env CB_AGG_GEMM_${1%%=*}=${1##*=} CB_AGG_GEMM_${2%%=*}=${2##*=} CB_AGG_GEMM_DBG=$d python tools/bench_agg_gemm.py --parts 0 2>&1 | tail -1 >> $O/agg_gemm_variants.txt; done
done
for ds in S-arxiv S-products; do python bench.py --dataset $ds --steps 10 --warmup 3 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_$ds.json 2>/dev/null; done
python bench.py --dataset S-cora --hip-graph 1 --steps 200 --warmup 20 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_S-cora_hipgraph.json 2>/dev/null
python bench.py --dataset S-pubmed --hip-graph 1 --layers 2 --steps 200 --warmup 20 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_S-pubmed_hipgraph.json 2>/dev/null
python tools/shard_probe.py --slices 4 > $O/shard_probe_S-pl10M.txt 2>&1
ls -la $O
