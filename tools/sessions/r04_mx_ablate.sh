export CLV_LSTM_MX=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | tail -5
for v in "" $MX_VARIANTS; do
  if [ -z "$v" ]; then unset CLV_LIB; echo "== base"; else export CLV_LIB=$PWD/abtest/mxabl$v/libclvae_hip.so; echo "== MX_ABL=$v"; fi
  timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy"
done
