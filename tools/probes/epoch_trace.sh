# Kernels of the reference epoch (bench.py --ref-epochs): aggregated by name over the LAST epoch -> gpurun_out/epoch_trace.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_ep -- python $R/bench.py --steps 2 --warmup 1 --ref-epochs 2 --cpu-baseline 0 --pmc-traffic 0 > /tmp/ep_line.json 2>/dev/null
DB=$(ls -t $(find /tmp/prof_ep -name "*.db") | head -1)
python - "$DB" > $R/gpurun_out/epoch_trace.txt <<'PY'
import sqlite3, sys, re, collections
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute('select name,start,end from kernels order by start'))
adam = [i for i, r in enumerate(rows) if 'k_adam' in r[0]]
# last epoch = from just after the second-to-last Adam launch up to the end of the trace
seq = rows[adam[-2] + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in seq:
    k = re.sub(r'\(.*', '', n).replace('void ', '').replace('cb::', '')[:90]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f'last epoch: {len(seq)} kernels, kernel time {tot/1e3:.1f} ms, span {(seq[-1][2]-seq[0][1])/1e6:.1f} ms')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{t/1e3:9.2f} ms  {c:4d}x  {k}')
PY
python -c "import json; d=json.load(open('/tmp/ep_line.json')); print({k: d[k] for k in d if 'epoch' in k or 'head_tail' in k})" >> $R/gpurun_out/epoch_trace.txt
