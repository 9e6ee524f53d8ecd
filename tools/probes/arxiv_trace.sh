# Per-kernel sequence of one S-arxiv step (eager) -> gpurun_out/trace_S-arxiv.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ds=${1:-S-arxiv}
rocprofv3 --kernel-trace -d /tmp/prof_$ds -- python $R/bench.py --dataset $ds --steps 10 --warmup 3 --cpu-baseline 0 --ref-epochs 0 --pmc-traffic 0 --dense-backward 0 > /dev/null 2>&1
DB=$(ls -t $(find /tmp/prof_$ds -name "*.db") | head -1)
python $R/tools/step_trace.py $DB > $R/gpurun_out/trace_$ds.txt
