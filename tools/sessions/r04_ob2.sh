#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "out_head or bernoulli or clip" 2>&1 | tail -3
for R in 32768 262144; do
  CLV_LIB=$PWD/abtest/obstamps/libclvae_hip.so R=$R python tools/out_head_bf16_stamps.py 2>&1 | grep -v amdgpu.ids
done
export TMPDIR=/tmp
for R in 32768 262144; do
  for F in 0 1; do
  (cd /tmp && R=$R CLV_OUT_HEAD_F32=$F rocprofv3 --kernel-trace --stats -d /tmp/prof_${R}_$F -o p --output-format csv -- python3 /root/repo/tools/head_bench.py > /dev/null 2>&1)
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/prof_${R}_$F/p_kernel_stats.csv')):
    if 'out_head' in r['Name'] or 'splitk' in r['Name']: print($R, r['Name'][:50], r['Calls'], r['AverageNs'])
PY
  done
done
