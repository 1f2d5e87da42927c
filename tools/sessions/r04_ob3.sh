#!/bin/bash
# end-to-end effect of out_head_bf16: bench cfg3 / cfg5, new kernel vs CLV_OUT_HEAD_F32=1, alternating
cd /root/repo
for i in 1 2; do
  for F in 0 1; do
    for W in cfg3 cfg5; do
      CLV_OUT_HEAD_F32=$F python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32=$F $W', d['ms_per_step'], d.get('block_ms'))"
    done
  done
done
