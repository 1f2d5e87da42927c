#!/bin/bash
cd /root/repo
for i in 1 2 3; do
  echo "== base";  python tools/mx_bench.py 2>&1 | grep -E "new_fwd" | head -3
  echo "== prio2"; CLV_LIB=/root/repo/abtest/mxpp/libclvae_hip.so python tools/mx_bench.py 2>&1 | grep -E "new_fwd" | head -3
done
