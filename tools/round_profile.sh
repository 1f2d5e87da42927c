#!/bin/bash
# Round-end evidence: the default bench line, a rocprofv3 kernel-stats summary of the same command, and the
# other workloads.  Run on the GPU box: bash tools/round_profile.sh <tag>
TAG=${1:-r01_c}; export TMPDIR=/tmp; R=$PWD
python bench.py > gpurun_out/${TAG}_bench_cfg3.json 2> gpurun_out/${TAG}_bench_cfg3.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_prof.log 2>&1)
cp gpurun_out/${TAG}_prof/p_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_cfg3.csv
for w in cfg5 cfg2; do python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; done
for w in gen1024 gen1; do python bench.py --workload $w --steps 2000 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err; done
tail -n 1 gpurun_out/${TAG}_bench_*.json | cut -c1-400
