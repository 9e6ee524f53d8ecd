#!/bin/bash
# kernel stats + per-dispatch step trace of the current tree (S-pl10M, 3 steps): usage  bash tools/probes/trace_now.sh <tag> ["ENV=.. ENV=.."] [extra bench flags]
TAG=${1:-now}; ENVS=$2; shift; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd /tmp && env $ENVS rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0 "$@" > $GRAFT_REPO_ROOT/$O/bench_line_under_rocprof.json 2>/dev/null )
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/bench_kernel_stats.md "$ENVS python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0 $*" > /dev/null 2>&1
python tools/step_trace.py $DB > $O/step_trace.txt 2>&1
