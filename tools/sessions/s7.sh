#!/bin/bash
# GPU session 7: where the wgrad stage time goes: ablated builds (wrong results by design)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s7; mkdir -p $O
for v in base abl1 abl2 abl3; do
  unset CLV_LIB; [ $v != base ] && export CLV_LIB=$R/abtest/$v/libclvae_hip.so
  echo "== $v"; python tools/wgrad_bench.py 262144 256 2>&1 | grep "nz= 0"; python tools/wgrad_bench.py 32768 128 2>&1 | grep "nz= 0"
done > $O/abl.log 2>&1
cat $O/abl.log
