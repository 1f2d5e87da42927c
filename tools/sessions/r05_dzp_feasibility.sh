#!/bin/bash
# Round 5: would dz written as three bf16 pieces by the backward kernels pay?  Two timing ablations (wrong results by design):
#  dzp_abl    lstm_pair.hip -DPAIR_ABL=10: the pair backward also splits every dz into pieces and issues three 2-byte stores
#  wb_nosplit wgrad_bf16.hip -DWB_ABLATE=1: the kernel-gradient product without the piece splitting (loads, LDS stores, MFMAs stay)
cd /root/repo; G=gpurun_out; O=$G/r05_dzp_feasibility.txt; : > $O
for i in 1 2 3; do
  for V in "" dzp_abl wb_nosplit; do
    L=""; [ -n "$V" ] && L=$PWD/abtest/$V/libclvae_hip.so
    CLV_LIB=$L python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-pmc-traffic --kernel-times 2>&1 >/dev/null | grep -E "lstm_pair_bwd|lstm_wgrad" | awk -v v="${V:-base}" '{print v, $1, $5}' | tr '\n' ' ' >> $O
    echo >> $O
  done
done
for V in "" wb_nosplit; do
  L=""; [ -n "$V" ] && L=$PWD/abtest/$V/libclvae_hip.so
  echo "== ${V:-base}" >> $O
  CLV_LIB=$L python tools/wgrad_bench.py 262144 256 2>/dev/null | cut -c1-60 >> $O
done
cat $O
