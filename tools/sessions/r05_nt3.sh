#!/bin/bash
# Round 5: the kernel-gradient product's operand loads nt: wbdz = dz (read once), wball = dz, h and x; config 5 and config 3
cd /root/repo; G=gpurun_out; O=$G/r05_nt3.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mx or wgrad" 2>&1 | grep -E "passed|failed" >> $O
for W in cfg5 cfg3; do
for i in 1 2 3; do
  for V in "" wbdz wball; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$W step, build %-6s' % ('$V' or 'base'), d['ms_per_step'])" >> $O
  done
done
done
cat $O
