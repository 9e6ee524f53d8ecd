#!/bin/bash
# Lists every kernel of the library that uses scratch memory or spills registers (hipcc -Rpass-analysis=kernel-resource-usage).
# A kernel of the hot path showing up here is a performance bug (see profiles/r02_gemm_presplit.md).  usage: bash tools/check_resources.sh
cd "$(dirname "$0")/../gnn-tail-generalization_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
name = None
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: name = m.group(1); continue
    m = re.search(r'(ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill): (\d+)', line)
    if m and int(m.group(2)) > 0: print('$f', name[:110], m.group(1), m.group(2))
"
done
