#!/bin/bash
# Round 5: the latent head with the transposed product (16-byte accesses) against the scalar layout (CLV_LATENT_V4=0), same build
cd /root/repo; G=gpurun_out; O=$G/r05_latent_v4.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "latent_head" 2>&1 | tail -3 >> $O
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "full_size or 32" 2>&1 | tail -3 >> $O
for i in 1 2 3; do
  for V in 1 0; do
    echo -n "CLV_LATENT_V4=$V  " >> $O
    CLV_LATENT_V4=$V python tools/latent_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
for i in 1 2 3; do
  for V in 1 0; do
    CLV_LATENT_V4=$V python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, CLV_LATENT_V4=$V', d['ms_per_step'])" >> $O
  done
done
cat $O
