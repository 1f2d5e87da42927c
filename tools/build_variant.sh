#!/bin/bash
# Build a variant of the library next to the in-tree one: abtest/<name>/libclvae_hip.so, with extra -D flags on the
# listed sources (the other objects are the in-tree ones).  abtest/ is git-ignored but travels with gpurun.
# Usage: bash tools/build_variant.sh <name> "<flags>" file1.hip [file2.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/classifying-vae-lstm_amd/csrc; O=$R/abtest/$NAME
mkdir -p $O
make -C $C -j4 > /dev/null
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value $FLAGS -c $f -o $O/$b.o
    OBJS="$OBJS $O/$b.o"
  else
    OBJS="$OBJS $C/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libclvae_hip.so $OBJS
echo built $O/libclvae_hip.so
