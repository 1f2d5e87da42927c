#!/bin/bash
# GPU session 11: cl_vae fused step with folded noise / loss means / counter; full suite
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s11; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --no-cpu-baseline --workload cfg2 2>&1 | cut -c1-260 > $O/cfg2.log; cat $O/cfg2.log
python bench.py --no-cpu-baseline --workload cfg2 --kernel-times 2>&1 | grep -v "^{" | cut -c1-200 > $O/cfg2_kt.log; cat $O/cfg2_kt.log
