#!/bin/bash
cd /root/repo
for i in 1 2 3; do
  for P in 1 0; do
    CLV_STAGE_IN_LABEL=$P python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stage_in_label=$P cfg3', d['ms_per_step'], d['final_loss'])"
  done
done
bash tools/kstats.sh st3 --workload cfg3 --no-also 2>&1 | grep -E "label_fwd|gather|sparse_proj|sum per"
