#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_timed_step.py tests/test_gpu_api.py tests/test_gpu_config1.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for i in 1 2 3; do
  for P in 1 0; do
    CLV_STAGE_IN_LABEL=$P python bench.py --workload cfg2 --no-also --no-cpu-baseline --no-roofline --steps 2000 --warmup 50 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stage=$P cfg2', d['ms_per_step'], d['final_loss'])"
  done
done
CLV_STAGE_IN_LABEL=1 python bench.py --workload cfg2 --bf16 --no-also --no-cpu-baseline --no-roofline --steps 2000 --warmup 50 2>/dev/null | tail -1 | cut -c1-140
