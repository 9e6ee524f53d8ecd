# does spacing the MFMAs of the multiplying wavefronts out (llc --amdgpu-mfma-padding-ratio: s_nops between neighbouring MFMAs) give the gathering
# wavefronts of the same SIMD their issue slots back?  Builds an alternate library on the box, runs the pair benchmark with both.
R=$GRAFT_REPO_ROOT
cd $R/gnn-tail-generalization_amd/csrc
for i in 1 2; do timeout 300 python $R/tools/bench_agg_gemm.py --iters 5 --parts 0 2>&1 | tail -1; done
cp ../lib/libcoldbrew_hip.so /tmp/lib_orig.so
for ratio in 50 100; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-cuda-compat -mllvm -amdgpu-mfma-padding-ratio=$ratio -c cb_agg_gemm.hip -o /tmp/cb_agg_gemm_pad.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcoldbrew_hip.so /tmp/cb_agg_gemm_pad.o $(ls ../_build/*.o | grep -v cb_agg_gemm.o)
  echo "--- padding ratio $ratio"
  for i in 1 2; do timeout 300 python $R/tools/bench_agg_gemm.py --iters 5 --parts 0 2>&1 | tail -1; done
done
cp /tmp/lib_orig.so ../lib/libcoldbrew_hip.so
