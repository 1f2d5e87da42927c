#!/bin/bash
# GPU session 20: new rows-per-workgroup rule: suite + config 5
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s20; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --kernel-times 2>&1 | cut -c1-170 > $O/cfg5.log; head -14 $O/cfg5.log; grep '"value"' $O/cfg5.log | cut -c1-200
