export CLV_LSTM_MX=1
for rep in 1 2 3; do
for v in "" $MX_VARIANTS; do
  if [ -z "$v" ]; then unset CLV_LIB; echo -n "== base   "; else export CLV_LIB=$PWD/abtest/$v/libclvae_hip.so; echo -n "== $v "; fi
  timeout 300 python tools/mx_bench.py 1024 256 32 2>&1 | grep -E "new_|copy" | tr '\n' ' '; echo
done; done
