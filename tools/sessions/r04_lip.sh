#!/bin/bash
cd /root/repo
for i in 1 2 3; do
  for P in 1 0; do
    CLV_LABEL_IN_PAIR=$P python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('label_in_pair=$P cfg3', d['ms_per_step'])"
  done
done
