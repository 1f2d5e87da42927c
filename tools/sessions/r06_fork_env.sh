#!/bin/bash
# Round 6: the forked front inside the captured step under the HIP runtime's graph switches
cd /root/repo; G=$PWD/gpurun_out; O=$G/r06_fork_env.txt; : > $O
run() { # label, env...
  L=$1; shift
  env "$@" python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 step  $L ', d['ms_per_step'])" >> $O
}
for i in 1 2; do
  run "serial" CLV_EXP_FORK=0
  run "fork" CLV_EXP_FORK=1
  run "fork QUEUES=2" CLV_EXP_FORK=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
  run "fork QUEUES=4" CLV_EXP_FORK=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
  run "fork PACKET_CAPTURE=0" CLV_EXP_FORK=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run "fork PACKET_CAPTURE=0 QUEUES=4" CLV_EXP_FORK=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
  run "serial PACKET_CAPTURE=0" CLV_EXP_FORK=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
cat $O
