#!/bin/bash
# GPU session 18: rows per workgroup of the VALU sequence kernels at 1024 rows
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s18; mkdir -p $O
for r in 2 4 1 2 4; do
  echo "== CLV_LSTM_ROWS=$r"
  CLV_LSTM_ROWS=$r timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 --kernel-times 2>&1 | grep -E "lstm_seq|\"value\"" | cut -c1-160
done > $O/rows.log 2>&1; cat $O/rows.log
