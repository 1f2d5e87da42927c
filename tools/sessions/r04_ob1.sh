#!/bin/bash
# out_head_bf16 first light: parity tests, then timing new vs f32 kernel (env switch) at cfg3 / cfg5 row counts
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "out_head or bernoulli or clip" 2>&1 | tail -8
for R in 32768 262144; do
  R=$R python tools/head_bench.py
  R=$R CLV_OUT_HEAD_F32=1 python tools/head_bench.py
done
