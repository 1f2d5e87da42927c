#!/bin/bash
# GPU session 19: rows per workgroup vs batch size
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s19; mkdir -p $O
for B in 384 512 768 1024 2048; do
  for r in 1 2 4; do
    CLV_LSTM_MFMA=0 CLV_LSTM_ROWS=$r timeout 120 python tools/lstm_rows_bench.py $B 128 2>&1 | grep "^B"
  done
  CLV_LSTM_MFMA=1 timeout 120 python tools/lstm_rows_bench.py $B 128 2>&1 | grep "^B"
done > $O/rows.log 2>&1; cat $O/rows.log
