#!/bin/bash
# GPU session 17: one-launch Adam-WN (grid barriers)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s17; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for m in 1 0 1 0; do
  echo "== CLV_ADAM_FUSED=$m"
  CLV_ADAM_FUSED=$m timeout 300 python bench.py --no-cpu-baseline --kernel-times 2>&1 | grep -E "adam|\"value\"" | cut -c1-160
done > $O/adam.log 2>&1; cat $O/adam.log
CLV_ADAM_FUSED=1 timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 5 --kernel-times 2>&1 | grep -E "adam|\"value\"" | cut -c1-160
