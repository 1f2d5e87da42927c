#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or mx or out_head" 2>&1 | tail -2
python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60
export TMPDIR=/tmp
for V in new obshift; do
for R in 32768 262144; do
  if [ $V = new ]; then unset CLV_LIB; else export CLV_LIB=/root/repo/abtest/$V/libclvae_hip.so; fi
  (cd /tmp && R=$R rocprofv3 --kernel-trace --stats -d /tmp/prof_${R}_$V -o p --output-format csv -- python3 /root/repo/tools/head_bench.py > /dev/null 2>&1)
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/prof_${R}_$V/p_kernel_stats.csv')):
    if 'out_head' in r['Name']: print('$V', $R, r['Name'][:50], r['Calls'], r['AverageNs'])
PY
done
done
