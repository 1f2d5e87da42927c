#!/bin/bash
# Round 6: would the hW kernel-gradient product (dense_outer_bf16) hide beside the slab reduction?  Two eager streams, kernel trace.
# (CLV_EXP_OUTER_FORK: a timing experiment in engine.py, not kept)
cd /root/repo; G=$PWD/gpurun_out
for V in 0 1; do
(cd /tmp && export TMPDIR=/tmp CLV_EXP_OUTER_FORK=$V && rocprofv3 --kernel-trace -d $G/r06_outer_fork_$V -o p --output-format csv -- python3 /root/repo/bench.py --workload ${1:-cfg3} --no-graph --steps 12 --warmup 3 --no-also --no-cpu-baseline --no-roofline --no-pmc-traffic > $G/r06_outer_fork.log 2>&1)
python3 - $V <<'PY'
import csv, glob, sys
f = glob.glob('/root/repo/gpurun_out/r06_outer_fork_%s/**/p_kernel_trace.csv' % sys.argv[1], recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('dense_outer', 'splitk_reduce_multi', 'lstm_wgrad', 'wn_fast_update'))]
sel = sel[-16:]
t0 = int(sel[0]['Start_Timestamp'])
print('CLV_EXP_OUTER_FORK=%s' % sys.argv[1])
for r in sel:
    print('  %-34s start %9.1f us  end %9.1f us  queue %s' % (r['Kernel_Name'].replace('void clv::', '').replace('clv::', '')[:34], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r.get('Queue_Id')))
PY
done
