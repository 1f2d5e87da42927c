#!/bin/bash
# the multi-GPU schedule on ONE GPU (no collective runs; CLV_FORCE_DP_GRAPHS=1) against the single-graph step, config 3
cd /root/repo
for i in 1 2 3; do
  python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('single graph      ', d['ms_per_step'], d.get('host_issue_us_per_step'))"
  CLV_FORCE_DP_GRAPHS=1 python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forced DP schedule', d['ms_per_step'], d.get('host_issue_us_per_step'), d.get('dp_schedule'))"
done
