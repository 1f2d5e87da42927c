#!/bin/bash
# Round 5: the kernel-gradient product with symmetric waves (WB_SYM, in-tree) against round 4's producer / consumer waves
# (abtest/wb_old: bash tools/build_variant.sh wb_old "-DWB_SYM=0 -DWB_SHARE=0" wgrad_bf16.hip).  Correctness first, then the
# product alone (tools/wgrad_bench.py), then the steps, alternating.
cd /root/repo; G=gpurun_out; V=$PWD/abtest/wb_old/libclvae_hip.so; O=$G/r05_wbsym.txt
python -m pytest tests/test_gpu_ops.py -q -x -k "wgrad or six_of_nine" 2>&1 | tail -3 > $O
for i in 1 2; do
  for L in "" $V; do
    echo "== $( [ -z "$L" ] && echo sym || echo old )" >> $O
    CLV_LIB=$L python tools/wgrad_bench.py 32768 128 2>/dev/null | cut -c1-80 >> $O
    CLV_LIB=$L python tools/wgrad_bench.py 262144 256 2>/dev/null | cut -c1-80 >> $O
  done
done
for i in 1 2 3; do
  for L in "" $V; do
    for W in cfg3 cfg5; do
      CLV_LIB=$L python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('old' if '$L' else 'sym', '$W', d['ms_per_step'])" >> $O
    done
  done
done
cat $O
