#!/bin/bash
# GPU session 12: bf16 variant of the fused cl_vae step
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s12; mkdir -p $O
python -m pytest tests/test_gpu_models.py -m gpu -q -x -s -k "bf16 or draws" > $O/pytest.log 2>&1; grep -E "cl_vae config 2|passed|failed|Error" $O/pytest.log | tail -8
python bench.py --no-cpu-baseline --workload cfg2 2>&1 | cut -c1-200 > $O/cfg2.log; cat $O/cfg2.log
python bench.py --no-cpu-baseline --workload cfg2 --bf16 2>&1 | cut -c1-200 > $O/cfg2_bf16.log; cat $O/cfg2_bf16.log
python bench.py --no-cpu-baseline --workload cfg2 --bf16 --kernel-times 2>&1 | grep -v "^{" | cut -c1-200
