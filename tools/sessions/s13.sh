#!/bin/bash
# GPU session 13: pair forward with k-pair packed FMAs
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s13; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "vrnn or pair or lstm or generation or golden" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py --no-cpu-baseline --kernel-times 2>&1 | cut -c1-170 > $O/cfg3.log; head -8 $O/cfg3.log; grep '"value"' $O/cfg3.log | cut -c1-150
python bench.py --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-200
