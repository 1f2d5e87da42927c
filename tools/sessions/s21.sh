#!/bin/bash
# GPU session 21: evidence files for profiles/ (stage timeline of the fused cl_vae step, MFMA probe)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s21; mkdir -p $O
CLV_LIB=$R/abtest/stamps/libclvae_hip.so python tools/vae_stamps.py > $O/vae_stamps.txt 2>/dev/null; cat $O/vae_stamps.txt
./tools/probes/mfma4x4_probe > $O/mfma4x4_probe.txt 2>&1; cat $O/mfma4x4_probe.txt
