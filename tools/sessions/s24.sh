#!/bin/bash
# GPU session 24: A/B of dZ inside the decoder's backward kernel (config 5)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s24; mkdir -p $O
for m in 1 0 1 0; do
  echo "== CLV_BWD_Z=$m"
  CLV_BWD_Z=$m timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --kernel-times 2>&1 | grep -E "lstm_seq_bwd|gemm_f32 |\"value\"" | cut -c1-160
done > $O/ab.log 2>&1; cat $O/ab.log
