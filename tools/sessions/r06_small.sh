#!/bin/bash
# Round 6: (1) byte gather with four rows per round (config 5), (2) the hW kernel-gradient product's wide form at configuration 3
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_small.txt; : > $O
python -m pytest tests/test_gpu_frames_u8.py tests/test_gpu_ops.py -q -x -k "gather or step_on_byte" 2>&1 | tail -3 >> $O
for i in 1 2 3; do
  for V in 200 100; do
    CLV_OUTER_WIDE_MIN=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, CLV_OUTER_WIDE_MIN=$V', d['ms_per_step'])" >> $O
  done
done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_small_prof -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg5 --steps 50 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_small_prof.log 2>&1)
python3 - <<'PY' >> $O
import csv
for r in list(csv.DictReader(open('/root/repo/gpurun_out/r06_small_prof/p_kernel_stats.csv')))[:18]:
    if 'gather' in r['Name'] or 'outer' in r['Name'] or 'window' in r['Name']:
        print('%-70s %5s calls  %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
for V in 200 100; do
(cd /tmp && CLV_OUTER_WIDE_MIN=$V TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_small_prof3_$V -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg3 --steps 100 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_small_prof3.log 2>&1)
python3 - $V <<'PY' >> $O
import csv, sys
for r in list(csv.DictReader(open('/root/repo/gpurun_out/r06_small_prof3_%s/p_kernel_stats.csv' % sys.argv[1])))[:14]:
    if 'outer' in r['Name']:
        print('WIDE_MIN=%s %-60s %5s calls  %9.1f us' % (sys.argv[1], r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $O
