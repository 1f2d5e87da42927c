#!/bin/bash
# Round 5: nt on the FORWARD pass's per-step stores only (mxf2; mxf2h0: the records nt, h default), three alternating rounds,
# kernels alone and the config-5 step
cd /root/repo; G=gpurun_out; O=$G/r05_mx_aux2.txt; : > $O
for i in 1 2 3; do
  for V in "" mxf2 mxf2h0; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    echo -n "== ${V:-base}  " >> $O
    timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' ' | sed 's/B 1024.*buffer)//' >> $O; echo >> $O
  done
done
for i in 1 2 3; do
  for V in "" mxf2 mxf2h0; do
    if [ -z "$V" ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$V/libclvae_hip.so; fi
    python bench.py --workload cfg5 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 step, build %-6s' % ('$V' or 'base'), d['ms_per_step'])" >> $O
  done
done
cat $O
