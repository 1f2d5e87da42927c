#!/bin/bash
# GPU session 16: z.Kz inside the MFMA sequence kernel (config 5)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s16; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -k "lstm_seq or full or vrnn" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for m in 1 0; do
  echo "== CLV_LSTM_MFMA=$m"
  CLV_LSTM_MFMA=$m python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 --kernel-times 2>&1 | grep -E "lstm_seq_fwd|gemm_f32 |sparse_proj|\"value\"" | cut -c1-160
done > $O/cfg5.log 2>&1; cat $O/cfg5.log
