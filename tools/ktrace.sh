export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-graph ${BENCH_ARGS} > $R/gpurun_out/kt.log 2>&1
head -40 $R/gpurun_out/kt/kt_kernel_stats.csv | cut -c1-160
