#!/bin/bash
# Round 5: the pair kernels' aux record [unit][kcarry, kc]: one 8-byte load per lane and step in the backward pass instead of two
# 4-byte ones.  auxold = the build before (abtest/auxold).
cd /root/repo; G=gpurun_out; O=$G/r05_pair_aux.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_timed_step.py -q -x -k "pair or vrnn or timed" 2>&1 | grep -E "passed|failed" >> $O
for i in 1 2 3; do
  for V in "" auxold; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-pmc-traffic --kernel-times 2>$G/kt.txt | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, build %-6s' % ('$V' or 'new'), d['ms_per_step'], 'pair launch us', d['roofline'].get('avg_launch_us'))" >> $O
    grep -E "lstm_pair_fwd|lstm_pair_bwd" $G/kt.txt | head -4 >> $O
  done
done
cat $O
