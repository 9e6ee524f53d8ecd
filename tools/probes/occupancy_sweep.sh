#!/bin/bash
# occupancy sensitivity of the aggregation row kernel (blocks of 256 threads; the LDS pad caps the resident blocks per CU)
for pad in 0 20000 26000 32000 40000 53000 80000; do
  for var in "" 16x16; do
    echo "== pad=$pad variant=${var:-16x8} gather=0"
    CB_SPMM_GATHER=0 CB_SPMM_LDS_PAD=$pad CB_SPMM_VARIANT=$var python tools/bench_spmm.py --iters 5 2>&1 | tail -1
  done
done
echo "== default (gather policy 2, no pad)"; python tools/bench_spmm.py --iters 5 2>&1 | tail -1
for pad in 32000 53000; do echo "== gather policy 2 pad=$pad"; CB_SPMM_LDS_PAD=$pad python tools/bench_spmm.py --iters 5 2>&1 | tail -1; done
