#!/bin/bash
cd /root/repo
for i in 1 2; do
for V in base wbprio3 wbprio1; do
  if [ $V = base ]; then unset CLV_LIB; else export CLV_LIB=/root/repo/abtest/$V/libclvae_hip.so; fi
  echo "== $V"; python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60 | head -2; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60 | tail -1
done
done
