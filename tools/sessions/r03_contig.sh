#!/bin/bash
# Round 3: forward pair kernel, global accesses in contiguous lane order (+ ds_bpermute) against the recurrence's
# (unit, k-slice) order; rocprofv3 kernel averages of both builds, alternating, in one session.
#   bash tools/build_variant.sh quad "-DPAIR_CONTIG=0 -fno-slp-vectorize" lstm_pair.hip
export TMPDIR=/tmp; R=$PWD; G=$R/gpurun_out/r03_contig; mkdir -p $G
for i in 1 2; do
  for v in quad contig; do
    L=$R/classifying-vae-lstm_amd/libclvae_hip.so; [ $v = quad ] && L=$R/abtest/quad/libclvae_hip.so
    (cd /tmp && CLV_LIB=$L rocprofv3 --kernel-trace --stats -d $G/${v}_$i -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 10 > $G/${v}_$i.log 2>&1)
    echo "== $v $i: $(python3 -c "import json,sys; print(json.loads(open('$G/${v}_$i.log').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)"
    grep -E "lstm_pair_(fwd|bwd)" $G/${v}_$i/p_kernel_stats.csv | cut -d, -f1,4 | cut -c1-90
  done
done
