#!/bin/bash
# Build a variant of the library next to the in-tree one: abtest/<name>/libclvae_hip.so, with extra flags on the
# listed sources (the other objects are the in-tree ones).  abtest/ is git-ignored but travels with gpurun.
# The compile command of a listed source is the MAKEFILE's own (make -n prints it: per-file flags included), with the
# extra flags appended through EXTRA and the object redirected; MAKEVARS passes variables to that make (e.g. NOSLP= to
# build a source WITHOUT its -fno-slp-vectorize).
# Usage: [MAKEVARS="NOSLP="] bash tools/build_variant.sh <name> "<flags>" file1.hip [file2.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/classifying-vae-lstm_amd/csrc; O=$R/abtest/$NAME
mkdir -p $O
make -C $C -j4 > /dev/null
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    CMD=$(make -C $C -n -B $b.o EXTRA="$FLAGS" $MAKEVARS | grep -- "-c $b.hip" | head -1)
    [ -n "$CMD" ] || { echo "no compile command for $b.hip"; exit 1; }
    CMD=${CMD/-o $b.o/-o $O/$b.o}
    echo "$CMD"
    (cd $C && eval "$CMD")
    OBJS="$OBJS $O/$b.o"
  else
    OBJS="$OBJS $C/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libclvae_hip.so $OBJS
echo built $O/libclvae_hip.so
