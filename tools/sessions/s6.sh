#!/bin/bash
# GPU session 6: wgrad producers on the fast path
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s6; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "wgrad or gemm" > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
python tools/wgrad_bench.py 2>&1 | grep -v amdgpu > $O/wgrad.log; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu >> $O/wgrad.log; cat $O/wgrad.log
python bench.py --no-cpu-baseline --kernel-times > $O/bench_new.json 2> $O/ktimes_new.txt
grep -v amdgpu.ids $O/ktimes_new.txt | head -12; cut -c1-120 $O/bench_new.json
