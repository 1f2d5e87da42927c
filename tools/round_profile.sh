#!/bin/bash
# Round evidence on the GPU box: bash tools/round_profile.sh <tag>   (e.g. r02_b) -> gpurun_out/<tag>_*; copy to profiles/.
#  1. the default bench line (and the other workloads)
#  2. rocprofv3 --kernel-trace --stats of the same command
#  3. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
#  4. SQ counters (VALU / MFMA busy, LDS conflicts) in two more passes (config 2 too, since round 5)
# PMC passes never carry another trace domain; the program after `--` is python3 itself.
TAG=${1:-r02}; export TMPDIR=/tmp; R=$PWD; G=$R/gpurun_out; mkdir -p $G
python bench.py > $G/${TAG}_bench_cfg3.json 2> $G/${TAG}_bench_cfg3.err
python bench.py --no-cpu-baseline --no-also --kernel-times > /dev/null 2> $G/${TAG}_kernel_times_cfg3.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d $G/${TAG}_prof -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-also --no-pmc-traffic > $G/${TAG}_prof.log 2>&1)
cp $G/${TAG}_prof/p_kernel_stats.csv $G/${TAG}_kernel_stats_cfg3.csv
for w in cfg5 cfg2; do
  python bench.py --workload $w --steps 200 --warmup 10 > $G/${TAG}_bench_$w.json 2> $G/${TAG}_bench_$w.err
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $G/${TAG}_prof_$w -o p --output-format csv -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-pmc-traffic > $G/${TAG}_prof_$w.log 2>&1)
  cp $G/${TAG}_prof_$w/p_kernel_stats.csv $G/${TAG}_kernel_stats_$w.csv
done
python bench.py --workload cfg2 --bf16 --steps 200 --warmup 10 --no-cpu-baseline > $G/${TAG}_bench_cfg2_bf16.json 2> $G/${TAG}_bench_cfg2_bf16.err
for w in gen1024 gen1 gen_vae1024 gen_vae1; do python bench.py --workload $w --steps 2000 --warmup 10 --no-cpu-baseline > $G/${TAG}_bench_$w.json 2> $G/${TAG}_bench_$w.err; done
for w in cfg3 cfg5 cfg2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $G/${TAG}_pmc_${c}_$w -o t --output-format csv -- python3 $R/bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $G/${TAG}_pmc_${c}_$w.log 2>&1)
  done
  python3 tools/pmc_traffic.py $w $G/${TAG}_pmc_FETCH_SIZE_$w/t_counter_collection.csv $G/${TAG}_pmc_WRITE_SIZE_$w/t_counter_collection.csv $G/${TAG}_pmc_traffic_$w.json 9
done
for w in cfg3 cfg5 cfg2; do
  (cd /tmp && rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $G/${TAG}_sq1_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $G/${TAG}_sq1_$w.log 2>&1)
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace -d $G/${TAG}_sq2_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $G/${TAG}_sq2_$w.log 2>&1)
  python3 tools/pmc_sq.py $G/${TAG}_sq_$w.json $G/${TAG}_sq1_$w/s_counter_collection.csv $G/${TAG}_sq2_$w/s_counter_collection.csv
done
tail -n 1 $G/${TAG}_bench_*.json | cut -c1-300
head -12 $G/${TAG}_kernel_stats_cfg3.csv | cut -c1-150
