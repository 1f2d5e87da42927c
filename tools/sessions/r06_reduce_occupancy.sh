cd /root/repo; G=$PWD/gpurun_out
for W in cfg3 cfg5; do for L in 0 24576 49152 65536; do
  S=100; [ $W = cfg5 ] && S=50
  (cd /tmp && CLV_LIB=/root/repo/abtest/redlds/libclvae_hip.so CLV_EXP_REDUCE_LDS=$L TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $G/kstats_rl -o p --output-format csv -- python3 /root/repo/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-also > $G/kstats_rl.log 2>&1)
  python3 - $W $L <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/root/repo/gpurun_out/kstats_rl/p_kernel_stats.csv')))
r = [x for x in rows if 'splitk_reduce_multi' in x['Name']][0]
print('%s  extra LDS %6s B -> reduce launch %6.1f us' % (sys.argv[1], sys.argv[2], float(r['AverageNs']) / 1e3))
PY
done; done
