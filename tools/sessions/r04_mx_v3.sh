export CLV_LSTM_MX=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | tail -5
timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy|old_" | tr '\n' ' '; echo
CLV_LIB=$PWD/abtest/mxstamps/libclvae_hip.so python tools/mx_stamps.py 2>&1 | grep -v amdgpu.ids
CLV_LIB=$PWD/abtest/mxstamps/libclvae_hip.so python tools/mx_stamps.py z 2>&1 | grep -v amdgpu.ids
unset CLV_LSTM_MX; exit 0
timeout 300 python -m pytest tests/test_gpu_api.py -q -x -k "real_rccl" 2>&1 | grep -E "Error|error|assert|^E " | head -30
