#!/bin/bash
# Builds a measurement variant of the library into _scratch/lib_<tag>.so: bash tools/probes/build_variant.sh <tag> "<extra hipcc flags>" [files...]
# (objects of the files not listed are taken from the regular build).  Run a tool against it with CB_EXP_LIB=_scratch/lib_<tag>.so python tools/probes/ab_lib.py <tool> ...
set -e
cd "$(dirname "$0")/../.."
tag=$1; flags=$2; shift 2
files=${@:-cb_spmm.hip}
mkdir -p _scratch/build_$tag
objs=""
for f in gnn-tail-generalization_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if echo " $files " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-cuda-compat $flags -c $f -o _scratch/build_$tag/$b.o
    objs="$objs _scratch/build_$tag/$b.o"
  else
    objs="$objs gnn-tail-generalization_amd/_build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _scratch/lib_$tag.so $objs
echo built _scratch/lib_$tag.so
