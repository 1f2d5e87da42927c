#!/bin/bash
# Round 5: (a) the committed form of lstm_mx (forward records nt) against the same source with -DMX_FWD_AUX=0 (mxf0); (b) nt on
# every per-step store of the pair kernels (pairnt) at config 3
cd /root/repo; G=gpurun_out; O=$G/r05_nt.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | grep -E "passed|failed" >> $O
for i in 1 2 3; do
  for V in "" mxf0; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, build %-6s' % ('$V' or 'new'), d['ms_per_step'])" >> $O
  done
done
for i in 1 2 3; do
  for V in "" pairnt; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, build %-6s' % ('$V' or 'base'), d['ms_per_step'])" >> $O
  done
done
cat $O
