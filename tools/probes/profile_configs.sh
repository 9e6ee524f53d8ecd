# Bench line + rocprofv3 kernel stats of the other BASELINE shapes (synthetic stand-ins) and label propagation at arxiv scale.
# usage (GPU box): bash tools/probes/profile_configs.sh   -> gpurun_out/r02_cfg_*
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ds in S-arxiv S-products; do
  python $R/bench.py --dataset $ds --steps 10 --warmup 3 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $R/gpurun_out/r02_cfg_${ds}_bench_line.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$ds -- python $R/bench.py --dataset $ds --steps 5 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > /dev/null 2>&1
  DB=$(ls -t $(find /tmp/prof_$ds -name "*.db") | head -1)
  python $R/tools/prof_summary.py $DB $R/gpurun_out/r02_cfg_${ds}_kernel_stats.md "python bench.py --dataset $ds --steps 5 --warmup 2 --cpu-baseline 0 (N=1)" > /dev/null
done
for ds in S-cora S-pubmed; do
  python $R/bench.py --dataset $ds --steps 50 --warmup 10 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --hip-graph 1 > $R/gpurun_out/r02_cfg_${ds}_hipgraph_bench_line.json 2>/dev/null
done
python $R/tools/bench_lp.py > $R/gpurun_out/r02_cfg_lp_arxiv.txt 2>&1
