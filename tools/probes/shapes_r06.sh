#!/bin/bash
# Round-6 bench lines of the other BASELINE shapes and trunk forms (one box): usage bash tools/probes/shapes_r06.sh <tag>
TAG=${1:-r06s}
O=gpurun_out/$TAG
mkdir -p $O
B="--cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0"
python bench.py --steps 6 --warmup 2 --se 111 $B > $O/bench_line_S-pl10M_se111.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --agg-dtype bf16 $B > $O/bench_line_S-pl10M_bf16agg.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --extra "--force_set_to_best_config=0 --type_trick=Residual" $B > $O/bench_line_S-pl10M_Residual.json 2>/dev/null
python bench.py --steps 6 --warmup 2 --extra "--force_set_to_best_config=0 --type_trick=NoResNodeNorm" $B > $O/bench_line_S-pl10M_NoResNodeNorm.json 2>/dev/null
python bench.py --dataset S-products --steps 10 --warmup 3 $B > $O/bench_line_S-products.json 2>/dev/null
python bench.py --dataset S-arxiv --steps 50 --warmup 10 $B > $O/bench_line_S-arxiv.json 2>/dev/null
python bench.py --dataset S-pubmed --se 111 --layers 2 --agg-dtype bf16 --hip-graph 1 --steps 200 --warmup 20 $B > $O/bench_line_S-pubmed_config2_hipgraph.json 2>/dev/null
python bench.py --dataset S-cora --hip-graph 1 --steps 300 --warmup 30 $B > $O/bench_line_S-cora_hipgraph.json 2>/dev/null
for f in $O/bench_line_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print(f"{sys.argv[1].split('/')[-1]:55s} {d['ms_per_step']:9.3f} ms/step  all-rows-fwd {d.get('all_rows_forward', {}).get('ms_per_step')}  dense {d.get('dense_backward', {}).get('ms_per_step')}  loss {d.get('final_loss')}  roofline {r.get('frac')}  peak {d.get('peak_mem_gb')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
