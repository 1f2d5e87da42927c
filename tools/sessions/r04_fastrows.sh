#!/bin/bash
cd /root/repo
for V in base fr48 fr32; do
  if [ $V = base ]; then unset CLV_LIB; else export CLV_LIB=/root/repo/abtest/$V/libclvae_hip.so; fi
  python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "adam or fast" 2>&1 | tail -1 | sed "s/^/$V /"
  bash tools/kstats.sh fr_$V --workload cfg3 --no-also 2>&1 | grep -E "wn_fast|sum per" | sed "s/^/$V /"
done
