#!/bin/bash
# GPU session 26: pair forward without the per-lane 64-bit row products (A/B against abtest/old)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s26; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "vrnn or pair or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in new old new old; do
  unset CLV_LIB; [ $v = old ] && export CLV_LIB=$R/abtest/old/libclvae_hip.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --kernel-times 2>&1 | grep -E "lstm_pair|\"value\"" | cut -c1-150
done > $O/ab.log 2>&1; cat $O/ab.log
