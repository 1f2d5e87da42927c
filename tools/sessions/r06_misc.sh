#!/bin/bash
# Round 6: (1) full GPU suite on the tree, (2) the DP schedule's cost on one GPU, (3) the cl_vae stage timeline of this round
cd /root/repo; G=$PWD/gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $G/r06_gpu_suite.txt
python tools/dp_overhead.py > $G/r06_dp_overhead.txt 2>&1
CLV_LIB=$PWD/abtest/stamps/libclvae_hip.so python tools/vae_stamps.py > $G/r06_vae_stage_timeline.txt 2>&1
for i in 1 2 3; do python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step', d['ms_per_step'])" >> $G/r06_gpu_suite.txt; done
cat $G/r06_gpu_suite.txt $G/r06_dp_overhead.txt $G/r06_vae_stage_timeline.txt
