#!/bin/bash
# Round 6: the slab reduction launch -- 64 slab lanes for jobs with many slabs (CLV_REDUCE_Z64_MIN), the rider blocks' chunk (CLV_SR_KC builds)
cd /root/repo; G=$PWD/gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_timed_step.py tests/test_gpu_models.py -q -x 2>&1 | tail -2
ks() { # tag workload steps env...
  T=$1; W=$2; S=$3; shift 3
  (cd /tmp && env "$@" TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/kstats_$T -o p --output-format csv -- python3 /root/repo/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/kstats_$T.log 2>&1)
  python3 - $T <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open('/root/repo/gpurun_out/kstats_%s/p_kernel_stats.csv' % sys.argv[1])))
r = [x for x in rows if 'splitk_reduce_multi' in x['Name']][0]
ms = json.loads(open('/root/repo/gpurun_out/kstats_%s.log' % sys.argv[1]).read().strip().split('\n')[-1])['ms_per_step']
print('%-28s reduce launch %6.1f us   step %.4f ms' % (sys.argv[1], float(r['AverageNs']) / 1e3, ms))
PY
}
for W in cfg3 cfg5; do
  S=100; [ $W = cfg5 ] && S=50
  ks ${W}_z64_kc64 $W $S A=1
  ks ${W}_noz64_kc64 $W $S CLV_REDUCE_Z64_MIN=1000000
  ks ${W}_z64_kc32 $W $S CLV_LIB=$PWD/abtest/srkc32/libclvae_hip.so
  ks ${W}_z64_kc128 $W $S CLV_LIB=$PWD/abtest/srkc128/libclvae_hip.so
  ks ${W}_z64_kc64_nowide $W $S CLV_REDUCE_WIDE_MAX=0
done
