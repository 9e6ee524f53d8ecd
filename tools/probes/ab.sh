#!/bin/bash
# A/B of environment switches on ONE box, alternating (A B A B ...): usage  bash tools/probes/ab.sh ROUNDS "ENV_A" "ENV_B" ["ENV_C" ...]
# (box-to-box spread of the step time is ~1.5 %: only alternating runs on the same box separate effects of a millisecond)
R=$GRAFT_REPO_ROOT
cd $R
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    ms=$(env $e timeout 600 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f %s' % (d['ms_per_step'], d.get('final_loss')))")
    echo "round $r  [$e]  $ms"
  done
done
