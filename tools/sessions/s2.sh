#!/bin/bash
# GPU session 2: new HEAD (single barrier per step + select/stride fixes) per-kernel numbers, prefetch depth 3.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do
  for v in new pf3 r01; do
    if [ $v = new ]; then unset CLV_LIB; else export CLV_LIB=$R/abtest/$v/libclvae_hip.so; fi
    echo "== $v fixed-cost" ; python tools/pair_fixed_cost.py 2>&1 | grep "T="
  done
done > $O/fixed.log 2>&1
for i in 1 2 3; do
  for v in new pf3 r01; do
    unset CLV_LIB
    case $v in r01|pf3) export CLV_LIB=$R/abtest/$v/libclvae_hip.so;; esac
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done > $O/ab.log 2>&1
unset CLV_LIB
python bench.py --no-cpu-baseline --kernel-times > $O/bench_new.json 2> $O/ktimes_new.txt
python bench.py --no-cpu-baseline --workload cfg5 --kernel-times > $O/bench_cfg5.json 2> $O/ktimes_cfg5.txt
python bench.py --no-cpu-baseline --workload cfg2 --kernel-times > $O/bench_cfg2.json 2> $O/ktimes_cfg2.txt
cat $O/fixed.log $O/ab.log; grep -v amdgpu.ids $O/ktimes_new.txt | head -24; grep -v amdgpu.ids $O/ktimes_cfg5.txt | head -24; grep -v amdgpu.ids $O/ktimes_cfg2.txt | head; cat $O/bench_cfg5.json $O/bench_cfg2.json | cut -c1-300
