#!/bin/bash
# rocprofv3 kernel stats of a workload's replayed step: bash tools/sessions/r06_kstats.sh <tag> <workload> [steps]
cd /root/repo; G=$PWD/gpurun_out; T=$1; W=$2; S=${3:-100}
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/kstats_$T -o p --output-format csv -- python3 /root/repo/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/kstats_$T.log 2>&1)
python3 - $T <<'PY'
import csv, sys
for r in list(csv.DictReader(open('/root/repo/gpurun_out/kstats_%s/p_kernel_stats.csv' % sys.argv[1])))[:16]:
    print('%-8s %-64s %5s calls  %9.1f us' % (sys.argv[1], r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -1 $G/kstats_$T.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
