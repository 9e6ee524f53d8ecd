#!/bin/bash
# Round-3 evidence run: bench line, rocprofv3 kernel stats + per-dispatch step trace of the same command, fused-kernel A/B, top-K, probes.
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python bench.py > $O/bench_default.json 2> $O/bench_default.err
CB_AGG_GEMM=0 python bench.py --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_two_kernel_form.json 2> /dev/null
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r03 -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > /dev/null 2>&1 )
DB=$(find /tmp/prof_r03 -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 (S-pl10M, N=1, round-3 tree, aggregation + GEMM kernels on)" > /dev/null 2>&1
python tools/step_trace.py $DB > $O/step_trace.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_topk -o tk -- python $GRAFT_REPO_ROOT/tools/bench_topk.py > $GRAFT_REPO_ROOT/$O/topk_line.txt 2>&1 )
python tools/prof_summary.py $(find /tmp/prof_topk -name "*.db" | head -1) $O/topk_kernel_stats.md "python tools/bench_topk.py (se_topk_replace, B = N = 169343, D = 768, K = 2: ogbn-arxiv scale)" > /dev/null 2>&1
python tools/bench_agg_gemm.py > $O/agg_gemm_parts.txt 2>&1
for v in "V=1" "NG=4 U=16" "NG=8 U=16" "NG=8 U=8"; do
  set -- $v
  if [ "$1" = "V=1" ]; then echo "two-phase blocks (version 1)" >> $O/agg_gemm_variants.txt; CB_AGG_GEMM_V=1 python tools/bench_agg_gemm.py --parts 0 2>&1 | tail -1 >> $O/agg_gemm_variants.txt
  else echo "specialised, $1 $2" >> $O/agg_gemm_variants.txt; env CB_AGG_GEMM_${1%%=*}=${1##*=} CB_AGG_GEMM_${2%%=*}=${2##*=} python tools/bench_agg_gemm.py --parts 0 2>&1 | tail -1 >> $O/agg_gemm_variants.txt
       for d in 8 16 1 2 3; do echo "  dbg=$d" >> $O/agg_gemm_variants.txt; env CB_AGG_GEMM_${1%%=*}=${1##*=} CB_AGG_GEMM_${2%%=*}=${2##*=} CB_AGG_GEMM_DBG=$d python tools/bench_agg_gemm.py --parts 0 2>&1 | tail -1 >> $O/agg_gemm_variants.txt; done
  fi
done
python tools/shard_probe.py --slices 4 > $O/shard_probe_S-pl10M.txt 2>&1
COLDBREW_SLICE_WEIGHTS=1,1,1,1 python tools/shard_probe.py --slices 4 --worlds 8 > $O/shard_probe_S-pl10M_uniform_slices.txt 2>&1
python tools/shard_probe.py --name S-products --slices 4 --dense-ms 30 > $O/shard_probe_S-products.txt 2>&1
for ds in S-arxiv S-products; do CB_AGG_GEMM=1 python bench.py --dataset $ds --steps 10 --warmup 3 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_$ds.json 2>/dev/null; done
python bench.py --dataset S-cora --hip-graph 1 --steps 200 --warmup 20 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_S-cora_hipgraph.json 2>/dev/null
python bench.py --dataset S-pubmed --hip-graph 1 --layers 2 --steps 200 --warmup 20 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $O/bench_S-pubmed_hipgraph.json 2>/dev/null
ls -la $O
