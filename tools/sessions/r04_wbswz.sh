#!/bin/bash
cd /root/repo
V=/root/repo/abtest/wbswz/libclvae_hip.so
CLV_LIB=$V python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
for i in 1 2; do
  echo "== linear pitch"; python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60 | head -2; python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
  echo "== swizzled";  CLV_LIB=$V python tools/wgrad_bench.py 32768 128 2>&1 | grep -v amdgpu | cut -c1-60 | head -2; CLV_LIB=$V python tools/wgrad_bench.py 262144 256 2>&1 | grep -v amdgpu | cut -c1-60
done
export TMPDIR=/tmp
for L in base swz; do
  if [ $L = swz ]; then export CLV_LIB=$V; else unset CLV_LIB; fi
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d /tmp/sq_$L -o s --output-format csv -- python3 /root/repo/tools/wgrad_bench.py 32768 128 > /dev/null 2>&1)
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open('/tmp/sq_$L/s_counter_collection.csv')):
    if 'wgrad_bf16' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:40]][r['Counter_Name']] += float(r['Counter_Value']); n[r['Kernel_Name'][:40]] += 1
for k,v in acc.items():
    print('$L', k, 'conflict cycles / LDS active cycles = %.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'],1)), {c: int(x) for c,x in v.items()})
PY
done
