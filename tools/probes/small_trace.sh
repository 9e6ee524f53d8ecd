R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ds in S-cora S-pubmed; do
  rocprofv3 --kernel-trace -d /tmp/prof_$ds -- python $R/bench.py --dataset $ds --steps 10 --warmup 3 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > /dev/null 2>&1
  DB=$(ls -t $(find /tmp/prof_$ds -name "*.db") | head -1)
  python $R/tools/step_trace.py $DB > $R/gpurun_out/small_trace_$ds.txt
  python $R/bench.py --dataset $ds --steps 50 --warmup 10 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --hip-graph 1 2>/dev/null | cut -c1-250
  python $R/bench.py --dataset $ds --steps 50 --warmup 10 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | cut -c1-250
done
