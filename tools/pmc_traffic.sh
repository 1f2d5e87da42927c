#!/bin/bash
# HBM traffic per kernel launch (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes, no other trace domains.  Usage: bash tools/pmc_traffic.sh [workload]   (on the GPU box)
export TMPDIR=/tmp; R=$PWD; W=${1:-cfg3}; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_$c -o t --output-format csv -- python3 $R/bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R && python3 tools/pmc_traffic.py $W gpurun_out/pmc_FETCH_SIZE/t_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/t_counter_collection.csv gpurun_out/r01_pmc_traffic_$W.json
