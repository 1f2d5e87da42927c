#!/bin/bash
# Round 6: does the frame projection launch overlap with the label launch when it is a parallel branch of the step's graph?
# (timing only: CLV_EXP_FORK=1 reads the previous step's byte batch)
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_fork.txt; : > $O
for i in 1 2 3; do
  for V in 0 1; do
    CLV_EXP_FORK=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, CLV_EXP_FORK=$V', d['ms_per_step'])" >> $O
  done
done
cat $O
