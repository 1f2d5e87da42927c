#!/bin/bash
# GPU session 27: final check of the committed tree: full suite, smoke, default bench line
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s27; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
