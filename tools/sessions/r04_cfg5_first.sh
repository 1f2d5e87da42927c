timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "full_size" 2>&1 | tail -5
for mx in 1 0; do
  echo "== CLV_USE_MX=$mx"
  CLV_USE_MX=$mx timeout 600 python bench.py --workload cfg5 --steps 40 --warmup 10 --no-cpu-baseline --kernel-times 2>gpurun_out/kt_cfg5_mx$mx.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  sort -t' ' -k2 gpurun_out/kt_cfg5_mx$mx.txt | head -40
done
