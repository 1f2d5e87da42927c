#!/bin/bash
# GPU session 8: label kernels (one-round-trip matvecs), Adam unit size
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s8; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for v in base u32 u64; do
  unset CLV_LIB; [ $v != base ] && export CLV_LIB=$R/abtest/$v/libclvae_hip.so
  echo "== $v"; python bench.py --no-cpu-baseline --kernel-times 2>&1 | grep -E "adam|label|value" | cut -c1-150
  python bench.py --no-cpu-baseline --workload cfg5 --kernel-times 2>&1 | grep -E "adam|\"value" | cut -c1-120
done > $O/adam.log 2>&1
cat $O/adam.log
