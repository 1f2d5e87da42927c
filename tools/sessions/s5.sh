#!/bin/bash
# GPU session 5: wgrad kernel with two stages of prefetch + the 16x16-lane slab reduction
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s5; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
python tools/wgrad_bench.py 2>&1 | grep -v amdgpu > $O/wgrad.log; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu >> $O/wgrad.log; cat $O/wgrad.log
for i in 1 2; do
  for v in new f32; do
    unset CLV_BF16_WGRAD; [ $v = f32 ] && export CLV_BF16_WGRAD=0
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done > $O/ab.log 2>&1
unset CLV_BF16_WGRAD
python bench.py --no-cpu-baseline --kernel-times > $O/bench_new.json 2> $O/ktimes_new.txt
cat $O/ab.log; grep -v amdgpu.ids $O/ktimes_new.txt | head -12
