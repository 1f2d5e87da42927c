#!/bin/bash
# GPU session 4: the split-bf16 weight-gradient kernel: correctness, then A/B against the f32 grouped GEMM
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s4; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -k "lstm_wgrad" > $O/pytest_wgrad.log 2>&1; tail -15 $O/pytest_wgrad.log
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
for i in 1 2; do
  for v in new f32; do
    unset CLV_BF16_WGRAD; [ $v = f32 ] && export CLV_BF16_WGRAD=0
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done > $O/ab.log 2>&1
unset CLV_BF16_WGRAD
python bench.py --no-cpu-baseline --kernel-times > $O/bench_new.json 2> $O/ktimes_new.txt
python bench.py --no-cpu-baseline --workload cfg5 --kernel-times > $O/bench_cfg5.json 2> $O/ktimes_cfg5.txt
cat $O/ab.log; grep -v amdgpu.ids $O/ktimes_new.txt | head -16; grep -v amdgpu.ids $O/ktimes_cfg5.txt | head -12; cut -c1-200 $O/bench_cfg5.json
