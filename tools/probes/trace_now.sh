#!/bin/bash
# kernel stats + per-dispatch step trace of the current tree (S-pl10M, 3 steps)
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_r03b -o r03b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 > $GRAFT_REPO_ROOT/$O/bench_line_under_rocprof.json 2>/dev/null )
DB=$(find /tmp/prof_r03b -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 (S-pl10M, N=1, flag hand-over + trunk backward tail)" > /dev/null 2>&1
python tools/step_trace.py $DB > $O/step_trace.txt 2>&1
