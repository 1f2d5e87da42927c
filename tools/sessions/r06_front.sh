#!/bin/bash
# Round 6: the frame projections inside the label launch (vrnn_front_kernel) against the two launches (CLV_FRONT_FUSED=0)
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_front.txt; : > $O
python -m pytest tests/test_gpu_front.py tests/test_gpu_timed_step.py tests/test_gpu_switches.py -q -x 2>&1 | tail -5 >> $O
for i in 1 2 3; do
  for V in 1 0; do
    CLV_FRONT_FUSED=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, CLV_FRONT_FUSED=$V', d['ms_per_step'])" >> $O
  done
done
for V in 1 0; do
(cd /tmp && CLV_FRONT_FUSED=$V TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_front_prof_$V -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg3 --steps 100 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_front_prof.log 2>&1)
python3 - $V <<'PY' >> $O
import csv, sys
for r in list(csv.DictReader(open('/root/repo/gpurun_out/r06_front_prof_%s/p_kernel_stats.csv' % sys.argv[1])))[:12]:
    print('FRONT_FUSED=%s %-60s %5s calls  %9.1f us' % (sys.argv[1], r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $O
