#!/bin/bash
# GPU session 1 (round 2): parity of the two-phase pair kernels, then A/B against the round-1 build.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
  for v in r01 new k10 k14; do
    if [ $v = new ]; then unset CLV_LIB; else export CLV_LIB=$R/abtest/$v/libclvae_hip.so; fi
    echo "== $v fixed-cost" ; python tools/pair_fixed_cost.py 2>&1 | grep "T="
  done
done > $O/fixed.log 2>&1
for i in 1 2 3; do
  for v in r01 new k10 k14 old1; do
    unset CLV_LIB CLV_PAIR_PHASES
    case $v in r01|k10|k14) export CLV_LIB=$R/abtest/$v/libclvae_hip.so;; old1) export CLV_PAIR_PHASES=1;; esac
    echo -n "$v "; python bench.py --no-cpu-baseline --steps 200 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done > $O/ab.log 2>&1
unset CLV_LIB CLV_PAIR_PHASES
python bench.py --no-cpu-baseline --kernel-times > $O/bench_new.json 2> $O/ktimes_new.txt
cat $O/fixed.log $O/ab.log; cat $O/ktimes_new.txt | head -30
