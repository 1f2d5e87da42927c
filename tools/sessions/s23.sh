#!/bin/bash
# GPU session 23: dZ inside the decoder's backward kernel
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s23; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --workload cfg5 --steps 50 --warmup 5 --kernel-times 2>&1 | cut -c1-170 > $O/cfg5.log; head -8 $O/cfg5.log; grep '"value"' $O/cfg5.log | cut -c1-200
