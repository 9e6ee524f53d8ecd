#!/bin/bash
# rocprofv3 kernel stats of the other shapes / trunk forms on one box: usage bash tools/probes/stats_r05.sh <tag>
TAG=${1:-r05st}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
B="--cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0"
run() {  # name, bench flags...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $R/bench.py $B "$@" > $R/$O/bench_line_$name.json 2>/dev/null )
  local DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  python tools/prof_summary.py $DB $O/kernel_stats_$name.md "python bench.py $B $*" > /dev/null 2>&1
}
run S-products --dataset S-products --steps 3 --warmup 1
run S-pl10M_se111 --se 111 --steps 3 --warmup 1
run S-arxiv_Initial --dataset S-arxiv --steps 20 --warmup 3
run S-arxiv_Residual --dataset S-arxiv --steps 20 --warmup 3 --extra "--force_set_to_best_config=0 --type_trick=Residual"
run S-arxiv_NoRes --dataset S-arxiv --steps 20 --warmup 3 --extra "--force_set_to_best_config=0 --type_trick=NoResNodeNorm"
ls $O
