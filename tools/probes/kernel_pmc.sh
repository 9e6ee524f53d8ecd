#!/bin/bash
# PMC table of selected kernels of ANY command: instruction mix, matrix-core busy cycles, effective clock, L2-side traffic.
# usage: bash tools/probes/kernel_pmc.sh <tag> "<kernel name substrings>" -- <command with ABSOLUTE paths: it runs from /tmp>
# One rocprofv3 --pmc pass per counter group (with --kernel-trace only: the pool refuses other trace domains next to --pmc).
TAG=$1; FILT=$2; shift; shift; [ "$1" = "--" ] && shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
: > $O/pmc_raw.txt
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "no counter file for [$c] (command failed?)" >> $O/pmc_raw.txt; continue; }
  echo "--- $c" >> $O/pmc_raw.txt
  python $R/tools/pmc_summary.py $f $FILT >> $O/pmc_raw.txt 2>&1
done
rm -rf /tmp/pmc_${TAG}_t
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_${TAG}_t -- "$@" > /dev/null 2>&1
f=$(find /tmp/pmc_${TAG}_t -name "*kernel_stats.csv" | head -1)
echo "--- kernel stats (no counters)" >> $O/pmc_raw.txt
for k in $FILT; do grep "$k" $f | cut -c1-200 >> $O/pmc_raw.txt; done
cat $O/pmc_raw.txt
