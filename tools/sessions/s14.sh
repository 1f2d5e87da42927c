#!/bin/bash
# GPU session 14: Adam column sums in one round, loss means in the reduce launch; full suite
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s14; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py --no-cpu-baseline --kernel-times 2>&1 | cut -c1-170 > $O/cfg3.log; head -16 $O/cfg3.log
python bench.py --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-200
python bench.py --no-cpu-baseline --workload cfg5 2>&1 | grep '"value"' | cut -c1-200
