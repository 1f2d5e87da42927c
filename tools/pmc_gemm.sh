export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_g1 -o g --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_g1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --kernel-trace -d $R/gpurun_out/pmc_g2 -o g --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_g2.log 2>&1
ls $R/gpurun_out/pmc_g1 $R/gpurun_out/pmc_g2
