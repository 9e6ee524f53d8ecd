R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
( cd /tmp && CB_BWD_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_ov -o ov -- python $R/bench.py --steps 3 --warmup 2 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 2>/dev/null | tail -1 | cut -c1-200 )
DB=$(find /tmp/prof_ov -name "*.db" | head -1)
python tools/step_trace.py $DB 2>&1 | tail -75 | cut -c1-130
