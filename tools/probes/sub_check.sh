# sub-wave stream kernel: small-graph debug, parity tests, then widths on the ogbn-products shape (every run under its own timeout)
R=$GRAFT_REPO_ROOT
cd $R
timeout 100 python tools/probes/dbg_sub.py 2>&1 | grep "bad rows"
timeout 600 python -m pytest tests/test_gpu_graph_spmm.py tests/test_gpu_random.py -x -q 2>&1 | tail -3
for d in 20 32 40 48 64; do
  for e in "" "CB_SPMM_NO_SUB=1"; do
    echo "--- d=$d $e"
    env $e timeout 60 python tools/bench_spmm.py --name S-products --n 2449029 --d $d --iters 5 2>&1 | tail -1
  done
done
timeout 120 python tools/bench_lp.py 2>&1 | tail -4
