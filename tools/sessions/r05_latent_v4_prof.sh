#!/bin/bash
# in-step kernel times of the latent head, V4 on / off (rocprofv3 kernel trace of bench.py --workload cfg5)
cd /root/repo; export TMPDIR=/tmp; R=$PWD; G=$R/gpurun_out; O=$G/r05_latent_v4_prof.txt; : > $O
for V in 1 0 1 0; do
  export CLV_LATENT_V4=$V
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $G/lv4_$V -o p --output-format csv -- python3 $R/bench.py --workload cfg5 --steps 60 --warmup 5 --no-cpu-baseline --no-also --no-pmc-traffic --no-roofline > $G/lv4_$V.log 2>&1)
  echo "== CLV_LATENT_V4=$V" >> $O
  grep -E "latent_head|lstm_mx_fwd|lstm_mx_bwd" $G/lv4_$V/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 >> $O
  tail -1 $G/lv4_$V.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step (profiled)', d['ms_per_step'])" >> $O
done
cat $O
