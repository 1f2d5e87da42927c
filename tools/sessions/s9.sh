#!/bin/bash
# GPU session 9: cl_vae fused step (table-driven, prefetched weights)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s9; mkdir -p $O
python -m pytest tests/test_gpu_models.py tests/test_gpu_api.py -m gpu -q -k "vae" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py --no-cpu-baseline --workload cfg2 --kernel-times 2>&1 | cut -c1-200 > $O/cfg2.log; cat $O/cfg2.log
CLV_LIB=$R/abtest/stamps/libclvae_hip.so python tools/vae_stamps.py > $O/stamps.log 2>&1; cat $O/stamps.log
