#!/bin/bash
# Round 5: is -fno-slp-vectorize still needed for correctness (no: the inline-asm select is gone), and is it worth keeping for speed?
# Before: MAKEVARS="NOSLP=" bash tools/build_variant.sh slp_on "" lstm_pair.hip lstm_mx.hip wgrad_bf16.hip out_head_bf16.hip outer_bf16.hip
cd /root/repo; G=gpurun_out; V=$PWD/abtest/slp_on/libclvae_hip.so
(cd tools/probes && hipcc --offload-arch=gfx950 -O3 -o trans_hazard_asm trans_hazard_asm.hip 2>/dev/null && ./trans_hazard_asm) > $G/r05_slp.txt 2>&1
for L in "" $V; do
  echo "== CLV_LIB='$L' (empty: the in-tree build, -fno-slp-vectorize on five sources; else: SLP on everywhere)" >> $G/r05_slp.txt
  CLV_LIB=$L timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py tests/test_gpu_timed_step.py -q -x 2>&1 | tail -3 >> $G/r05_slp.txt
done
for i in 1 2 3; do
  for L in "" $V; do
    for W in cfg3 cfg5; do
      CLV_LIB=$L python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slp_on' if '$L' else 'noslp ', '$W', d['ms_per_step'])" >> $G/r05_slp.txt
    done
  done
done
cat $G/r05_slp.txt
