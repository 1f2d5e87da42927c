#!/bin/bash
# Round 6: kernel trace of the forked front (label launch on the main stream, projection launch on a side stream, eager): do they overlap?
cd /root/repo; G=$PWD/gpurun_out
(cd /tmp && export TMPDIR=/tmp CLV_EXP_FORK_TIME=1 CLV_EXP_FORK=1 && rocprofv3 --kernel-trace -d $G/r06_fork_trace -o p --output-format csv -- python3 /root/repo/bench.py --workload cfg3 --no-graph --steps 12 --warmup 3 --no-also --no-cpu-baseline --no-roofline --no-pmc-traffic > $G/r06_fork_trace.log 2>&1)
python3 - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/r06_fork_trace/**/p_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if 'sparse_proj' in r['Kernel_Name'] or 'label_fwd_x' in r['Kernel_Name'] or 'lstm_pair_fwd' in r['Kernel_Name']]
t0 = int(sel[-30]['Start_Timestamp'])
for r in sel[-30:]:
    print('%-40s start %9.1f us  end %9.1f us  stream/queue %s' % (r['Kernel_Name'][:40], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id')))
PY
