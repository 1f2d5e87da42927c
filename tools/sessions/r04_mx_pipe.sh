export CLV_LSTM_MX=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mx" 2>&1 | tail -3
for rep in 1 2 3; do
for v in "" mxprev; do
  if [ -z "$v" ]; then unset CLV_LIB; echo -n "== new    "; else export CLV_LIB=$PWD/abtest/$v/libclvae_hip.so; echo -n "== $v "; fi
  timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' '; echo
done; done
