#!/bin/bash
# Round 6: eps_Z drawn inside latent_head_fwd (configuration 5: no Philox launch) against the round-5 tree (abtest/r05 = git archive d5b6064)
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_eps.txt; : > $O
python -m pytest tests/test_gpu_ops.py -q -x -k "latent_head" 2>&1 | tail -3 >> $O
python -m pytest tests/test_gpu_timed_step.py tests/test_gpu_models.py -q -x 2>&1 | tail -3 >> $O
for i in 1 2 3; do
  for V in r05 new; do
    if [ $V = r05 ]; then D=$PWD/abtest/r05; else D=$PWD; fi
    (cd $D && python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, tree %-4s' % '$V', d['ms_per_step'], d['timed_blocks']['ms_per_step'])") >> $O
  done
done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/r06_eps_prof -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg5 --steps 50 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/r06_eps_prof.log 2>&1)
python3 - <<'PY' >> $O
import csv
rows = list(csv.DictReader(open('/root/repo/gpurun_out/r06_eps_prof/p_kernel_stats.csv')))
for r in rows[:22]:
    print('%-70s %5s calls  %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat $O
