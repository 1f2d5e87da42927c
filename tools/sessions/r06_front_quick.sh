#!/bin/bash
# quick look: workgroup timelines of the front launch's variants, its tests, three alternating pairs of the configuration-3 step
cd /root/repo
for n in "$@"; do echo == $n; CLV_LIB=$PWD/abtest/front_$n/libclvae_hip.so python tools/front_timeline.py 2>&1 | tail -5 | head -3; done
python -m pytest tests/test_gpu_front.py -q -x 2>&1 | tail -2
for i in 1 2 3; do for V in 1 0; do echo -n "FRONT_FUSED=$V "; CLV_FRONT_FUSED=$V python bench.py --workload cfg3 --no-also --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done
