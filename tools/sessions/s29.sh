#!/bin/bash
# GPU session 29: conflict-free dz slices in the single-LSTM backward kernel
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s29; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "lstm or vrnn or golden or generation" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for B in 256 512 1024; do timeout 120 python tools/lstm_rows_bench.py $B 128 2>&1 | grep "^B"; done
timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --kernel-times 2>&1 | grep -E "lstm_seq|\"value\"" | cut -c1-170
