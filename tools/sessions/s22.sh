#!/bin/bash
# GPU session 22: full suite with the added edge cases
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s22; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
