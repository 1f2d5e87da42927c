#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gather" 2>&1 | tail -2
python -m pytest tests/test_gpu_api.py tests/test_gpu_config1.py tests/test_gpu_timed_step.py -x -q -m gpu 2>&1 | tail -2
bash tools/kstats.sh g3 --workload cfg3 --no-also 2>&1 | grep -E "gather|sum per|^\{"
bash tools/kstats.sh g5 --workload cfg5 --no-also --steps 40 2>&1 | grep -E "gather|sum per"
