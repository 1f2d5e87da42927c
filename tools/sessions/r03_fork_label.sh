for i in 1 2 3; do
for f in 0 1; do
CLV_FORK_LABEL=$f python bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fork $f', d['ms_per_step'], d['value'])"
done; done
