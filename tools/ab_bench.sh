#!/bin/bash
# A/B two builds of the library inside ONE gpurun call (box-to-box variance is ~15 %): alternate N times.
# Usage: bash tools/ab_bench.sh path/to/old/libclvae_hip.so [N] [workload]   (the new one is the in-tree build)
OLD=$1; N=${2:-3}; W=${3:-cfg3}; NEW=classifying-vae-lstm_amd/libclvae_hip.so
cp $NEW /tmp/new.so
for i in $(seq $N); do
  cp $OLD $NEW; echo -n "old "; python bench.py --workload $W --no-cpu-baseline --no-roofline | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"
  cp /tmp/new.so $NEW; echo -n "new "; python bench.py --workload $W --no-cpu-baseline --no-roofline | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"
done
