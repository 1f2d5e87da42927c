#!/bin/bash
# Round 5: the driver's own N > 1 command lines (torch.distributed.run, one rank per "GPU") at N = 2, 4, 8 with every rank on
# the box's ONE GPU (CLV_BENCH_SHARE_GPU=1, gloo): not a scaling curve -- the ranks time-share one device -- but every
# world-size-dependent branch of bench.py / trainer.py / parallel.py (state broadcast, schedule trial, bucket bounds, the
# all-gathered block times, the JSON's dp fields) runs at the world sizes the driver will use.
cd /root/repo; G=gpurun_out; O=$G/r05_share_n.txt; : > $O
export CLV_BENCH_SHARE_GPU=1
for N in 2 4 8; do
  echo "== N=$N: python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) bench.py --gpus $N --steps 5 --warmup 2" >> $O
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
      bench.py --gpus $N --steps 5 --warmup 2 > $G/r05_share_n_$N.out 2> $G/r05_share_n_$N.err
  echo "rc=$?" >> $O
  tail -1 $G/r05_share_n_$N.out | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'backend', 'shared_device', 'ms_per_step_by_rank', 'config')
print(json.dumps({k: d[k] for k in keep if k in d}))
print('dp_schedule:', json.dumps(d.get('dp_schedule')))
print('allreduce_alone:', json.dumps(d.get('allreduce_alone')))" >> $O 2>&1
  grep -i "error\|traceback" $G/r05_share_n_$N.err | head -5 >> $O
done
cat $O
