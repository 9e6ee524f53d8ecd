#!/bin/bash
# Round 5: rank 0's share of a P-way sharded S-pl10M step with the row-sparse levels COMPACT in the rank's rows (dist.ShardedGraph.support_levels).
# Three probes (cover plan): the full graph (forward aggregations + the last, dense level of the backward), level 0 and level 1 of the backward.
out=${1:-gpurun_out/r05e}
mkdir -p $out
n1=${N1_MS:-155.4}
drest=${DREST_MS:-52.8}
python tools/shard_probe.py --worlds 1,2,4,8 --mode cover --n1-ms $n1 --dense-cover-ms $drest > $out/shard_probe_S-pl10M_full.txt 2>&1
python tools/shard_probe.py --worlds 2,4,8 --mode cover --n1-ms $n1 --dense-cover-ms $drest --support-level 0 --compact 1 > $out/shard_probe_S-pl10M_level0_compact.txt 2>&1
python tools/shard_probe.py --worlds 2,4,8 --mode cover --n1-ms $n1 --dense-cover-ms $drest --support-level 1 --compact 1 > $out/shard_probe_S-pl10M_level1_compact.txt 2>&1
tail -n 8 $out/shard_probe_S-pl10M_full.txt $out/shard_probe_S-pl10M_level0_compact.txt $out/shard_probe_S-pl10M_level1_compact.txt
