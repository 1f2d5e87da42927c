timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_config1.py tests/test_gpu_timed_step.py -x -q 2>&1 | tail -6
for w in cfg3 cfg2 cfg5; do
  python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['value'], d['host_issue_us_per_step'])"
done
