#!/bin/bash
# Round 5: what the producer's work (1), the global stores (2), the input gather (3) and the backward stores (64) cost in the
# FINAL lstm_mx kernels (round 4 measured them on an earlier version): upper bounds for "note lists made by a pre-pass" and
# "five floats per unit and step".  Before: for v in 1 2 3 64; do bash tools/build_variant.sh mxabl$v "-DMX_ABL=$v" lstm_mx.hip; done
cd /root/repo; G=gpurun_out; O=$G/r05_mx_ablate.txt; : > $O
for i in 1 2; do
  for v in "" ${MXV:-1 2 3 64}; do
    if [ -z "$v" ]; then unset CLV_LIB; echo "== base" >> $O; else export CLV_LIB=$PWD/abtest/mxabl$v/libclvae_hip.so; echo "== MX_ABL=$v" >> $O; fi
    timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' ' >> $O; echo >> $O
  done
done
cat $O
