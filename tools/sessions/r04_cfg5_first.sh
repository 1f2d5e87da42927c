for mx in 1 0 1 0; do
  echo "== CLV_USE_MX=$mx"
  CLV_USE_MX=$mx timeout 600 python bench.py --workload cfg5 --steps 40 --warmup 10 --no-cpu-baseline --kernel-times 2>gpurun_out/kt_cfg5_mx$mx.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
