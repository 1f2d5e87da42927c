#!/bin/bash
# GPU session 28: SQ counter summaries for config 5 and config 2 (two PMC passes each, kernel trace only)
export TMPDIR=/tmp; R=$PWD; G=$R/gpurun_out; TAG=r02_e
for w in cfg5 cfg2; do
  (cd /tmp && rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $G/${TAG}_sq1_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $G/${TAG}_sq1_$w.log 2>&1)
  (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --kernel-trace -d $G/${TAG}_sq2_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $G/${TAG}_sq2_$w.log 2>&1)
  python3 tools/pmc_sq.py $G/${TAG}_sq_$w.json $G/${TAG}_sq1_$w/s_counter_collection.csv $G/${TAG}_sq2_$w/s_counter_collection.csv
done
ls -la $G/${TAG}_sq_cfg5.json $G/${TAG}_sq_cfg2.json
