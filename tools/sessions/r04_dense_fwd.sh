#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dense_window or outer" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py tests/test_gpu_timed_step.py tests/test_gpu_config1.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  for P in 1 0; do
    for W in cfg3 cfg5; do
      CLV_DENSE_HW_FWD=$P python bench.py --workload $W --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('densefwd=$P $W', d['ms_per_step'])"
    done
  done
done
bash tools/kstats.sh dfwd --workload cfg3 --no-also 2>&1 | grep -E "label_fwd|dense_window|dense_outer|sum per"
bash tools/kstats.sh dfwd5 --workload cfg5 --no-also --steps 40 2>&1 | grep -E "label_fwd|dense_window|dense_outer|sum per"
