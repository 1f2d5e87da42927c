set -x
export CLV_LSTM_MX=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | tail -15
timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | tail -12
