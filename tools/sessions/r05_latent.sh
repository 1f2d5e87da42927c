#!/bin/bash
# Round 5: the latent head at configuration 5 (76 + 88 us for ~230 / 317 MB): LDS row stride 116 -> 100 floats (still 64
# different banks per A-operand read: 100 = 36 mod 64) puts the forward kernel at 79 KB, two workgroups per CU.
# Before: bash tools/build_variant.sh lh100b "-DZH_LD_V=100" latent_head.hip; ... lh100 "-DZH_LD_V=100 -DZH_FWD_CAP=512"; lh1024 "... -DZH_FWD_CAP=1024"
cd /root/repo; G=gpurun_out; O=$G/r05_latent.txt; : > $O
for i in 1 2; do
  for V in "" lh100b lh100 lh1024; do
    L=""; [ -n "$V" ] && L=$PWD/abtest/$V/libclvae_hip.so
    python tools/latent_bench.py $L >> $O 2>&1
  done
done
for i in 1 2; do
  for V in "" lh100 lh1024; do
    L=""; [ -n "$V" ] && L=$PWD/abtest/$V/libclvae_hip.so
    CLV_LIB=$L python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, build %-8s' % ('$V' or 'in-tree'), d['ms_per_step'])" >> $O
  done
done
cat $O
