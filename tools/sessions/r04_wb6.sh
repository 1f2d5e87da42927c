#!/bin/bash
cd /root/repo
V=/root/repo/abtest/wb6/libclvae_hip.so
CLV_LIB=$V python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
for i in 1 2; do
  echo "== nine"; python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
  echo "== six";  CLV_LIB=$V python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60; CLV_LIB=$V python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
done
