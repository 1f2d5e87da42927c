#!/bin/bash
cd /root/repo
for W in 256 128 192 64; do
  CLV_SP_WGS=$W bash tools/kstats.sh sp$W --workload cfg3 --no-also 2>&1 | grep -E "sparse_proj|sum per" | sed "s/^/wgs=$W /"
done
