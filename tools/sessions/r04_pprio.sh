#!/bin/bash
cd /root/repo
for i in 1 2 3; do
for V in base pprio1 pprio2 pprio3; do
  if [ $V = base ]; then unset CLV_LIB; else export CLV_LIB=/root/repo/abtest/$V/libclvae_hip.so; fi
  python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V cfg3', d['ms_per_step'])"
done
done
