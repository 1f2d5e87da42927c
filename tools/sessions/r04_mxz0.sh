#!/bin/bash
cd /root/repo
CLV_LIB=/root/repo/abtest/mxz0/libclvae_hip.so python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mx" 2>&1 | tail -1
for i in 1 2 3; do
  echo "== base";  python tools/mx_bench.py 2>&1 | grep -E "new_fwd" | head -3
  echo "== z in wave 0"; CLV_LIB=/root/repo/abtest/mxz0/libclvae_hip.so python tools/mx_bench.py 2>&1 | grep -E "new_fwd" | head -3
done
