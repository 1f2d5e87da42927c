#!/bin/bash
# Round 5: lstm_mx's per-step record unit-interleaved ([unit][ki,kf,kg,ko], [unit][kcarry,kc]): 3 stores per lane and step in the
# forward pass instead of 7, 3 loads in the backward pass instead of 7.  mxold = the build before, mxa3 = + the gather FMAs
# three slots earlier.  Before: see the commit message / PERFLOG R5.8.
cd /root/repo; G=gpurun_out; O=$G/r05_mx_wide.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | tail -3 >> $O
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_timed_step.py -q -x -k "full_size or timed or 32" 2>&1 | tail -3 >> $O
for i in 1 2 3; do
  for V in "" mxold mxa3; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    echo -n "== ${V:-new}  " >> $O
    timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' ' | sed 's/B 1024.*buffer)//' >> $O; echo >> $O
  done
done
for i in 1 2 3; do
  for V in "" mxold mxa3; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, build %-6s' % ('$V' or 'new'), d['ms_per_step'])" >> $O
  done
done
cat $O
