#!/bin/bash
# the one-instruction residual (v_dot2c_f32_bf16) in wgrad_bf16 / lstm_mx / out_head_bf16: parity, then new vs previous commit
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or mx or out_head" 2>&1 | tail -3
P=$PWD/abtest/prev/libclvae_hip.so
for i in 1 2; do
  echo "== new";  python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
  echo "== prev"; CLV_LIB=$P python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60; CLV_LIB=$P python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
done
for i in 1 2 3; do
  for L in new prev; do
    for W in cfg3 cfg5; do
      if [ $L = prev ]; then export CLV_LIB=$P; else unset CLV_LIB; fi
      python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L $W', d['ms_per_step'])"
    done
  done
done
