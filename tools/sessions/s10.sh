#!/bin/bash
# GPU session 10: timing ablations of the fused cl_vae step
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s10; mkdir -p $O
for v in base vab1 vab2 vab3 vab5; do
  unset CLV_LIB; [ $v != base ] && export CLV_LIB=$R/abtest/$v/libclvae_hip.so
  echo "== $v"; python bench.py --no-cpu-baseline --workload cfg2 --kernel-times 2>&1 | grep -E "vae_fused" | cut -c1-150
done > $O/ablate.log 2>&1
cat $O/ablate.log
