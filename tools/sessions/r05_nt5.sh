#!/bin/bash
# Round 5: lstm_mx_bwd's dz stores nt INSIDE the step (alone: +4-10 us per launch)
cd /root/repo; G=gpurun_out; O=$G/r05_nt5.txt; : > $O
for i in 1 2 3; do
  for V in "" mxb2; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, build %-6s' % ('$V' or 'base'), d['ms_per_step'])" >> $O
  done
done
cat $O
