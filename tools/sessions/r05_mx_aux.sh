#!/bin/bash
# Round 5: cache policy of lstm_mx's per-step stores (forward records + h, backward dz): default, nt (2), sc1 (16), nt sc1 (18), sc0 (1)
cd /root/repo; G=gpurun_out; O=$G/r05_mx_aux.txt; : > $O
for i in 1 2; do
  for V in "" mxaux2 mxaux16 mxaux18 mxaux1; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    echo -n "== ${V:-base}  " >> $O
    timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' ' | sed 's/B 1024.*buffer)//' >> $O; echo >> $O
  done
done
cat $O
