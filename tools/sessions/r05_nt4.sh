#!/bin/bash
# Round 5: config 3, last-use loads nt in the pair kernels: px = the forward kernel's input projections, pr = the backward
# kernel's coefficient records, pxr = both
cd /root/repo; G=gpurun_out; O=$G/r05_nt4.txt; : > $O
for i in 1 2 3 4; do
  for V in "" px pr pxr; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, build %-6s' % ('$V' or 'base'), d['ms_per_step'])" >> $O
  done
done
cat $O
