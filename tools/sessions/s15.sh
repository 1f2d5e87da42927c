#!/bin/bash
# GPU session 15: LSTM forward on the 4x4x1 MFMA for large batches (config 5)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s15; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "lstm_seq" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for m in 1 0 1 0; do
  echo "== CLV_LSTM_MFMA=$m"
  CLV_LSTM_MFMA=$m python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 --kernel-times 2>&1 | grep -E "lstm_seq_fwd|\"value\"" | cut -c1-160
done > $O/cfg5.log 2>&1; cat $O/cfg5.log
