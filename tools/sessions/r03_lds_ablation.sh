# What the ds_read_b128 of a pair-kernel step cost on its critical path: timing ablations (wrong results) of lstm_pair.hip
#   bash tools/build_variant.sh abl1 "-DPAIR_ABL=1 -fno-slp-vectorize" lstm_pair.hip     (half of every slice read)
#   bash tools/build_variant.sh abl2 "-DPAIR_ABL=2 -fno-slp-vectorize" lstm_pair.hip     (nothing read)
for v in ${VARIANTS:-base abl1 abl2 base abl1 abl2}; do
  if [ "$v" = base ]; then unset CLV_LIB; else export CLV_LIB=$PWD/abtest/$v/libclvae_hip.so; fi
  python bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu-baseline --kernel-times 2>/tmp/kt.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
  grep -E "lstm_pair_(fwd|bwd)" /tmp/kt.txt
done
