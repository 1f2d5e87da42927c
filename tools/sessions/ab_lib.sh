#!/bin/bash
# usage: ab_lib.sh <variant-lib-dir> <workload> <steps>
cd /root/repo
for i in 1 2 3; do
  echo -n "$2 in-tree "; python bench.py --workload $2 --steps $3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "$2 $1 "; CLV_LIB=$PWD/abtest/$1/libclvae_hip.so python bench.py --workload $2 --steps $3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
