#!/bin/bash
# Round 6: the pair path (configuration 3) on byte frames: the label launch leaves a uint8 batch, sparse_proj / wgrad / out_head /
# dense_outer read bytes.  CLV_FRAMES_U8=0 is the float route of the same tree.
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_u8pair.txt; : > $O
python -m pytest tests/test_gpu_frames_u8.py tests/test_gpu_switches.py -q -x 2>&1 | tail -5 >> $O
python -m pytest tests/test_gpu_timed_step.py tests/test_gpu_models.py tests/test_gpu_api.py tests/test_gpu_bench_dp.py -q -x 2>&1 | tail -3 >> $O
for i in 1 2 3 4; do
  for V in 0 1; do
    CLV_FRAMES_U8=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step, CLV_FRAMES_U8=$V', d['ms_per_step'], d['timed_blocks']['ms_per_step'])" >> $O
  done
done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_u8pair_prof -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg3 --steps 100 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_u8pair_prof.log 2>&1)
python3 - <<'PY' >> $O
import csv
rows = list(csv.DictReader(open('/root/repo/gpurun_out/r06_u8pair_prof/p_kernel_stats.csv')))
for r in rows[:16]:
    print('%-70s %5s calls  %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat $O
