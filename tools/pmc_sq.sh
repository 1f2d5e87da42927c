export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_sq2 -o sq --output-format csv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc_sq2.log 2>&1
ls $R/gpurun_out/pmc_sq2/* | head
