#!/bin/bash
# Round-6 evidence on one box: the default bench run (driver flags; with cpu_baseline + PMC traffic), kernel stats + step trace under rocprofv3.
# usage: bash tools/probes/collect_r06.sh <tag>
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_S-pl10M_default_full_run.json 2> $O/bench_default.err
rm -rf /tmp/prof_$TAG
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0 > $R/$O/bench_line_under_rocprof.json 2>/dev/null )
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB $O/bench_kernel_stats.md "python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0 (S-pl10M, row-sparse backward)" > /dev/null 2>&1
python tools/step_trace.py $DB > $O/step_trace.txt 2>&1
python - "$O/bench_line_S-pl10M_default_full_run.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['final_loss'], d['roofline']['frac'], d['roofline']['traffic'], d.get('dense_backward', {}).get('ms_per_step'), d.get('dense_backward', {}).get('final_loss'), d.get('reference_epoch_ms'))
PY
