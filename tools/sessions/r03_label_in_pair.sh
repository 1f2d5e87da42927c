# the label path's backward as the pair backward kernel's epilogue (default) against its own launch (CLV_LABEL_IN_PAIR=0)
for i in 1 2 3; do for f in 0 1; do
CLV_LABEL_IN_PAIR=$f python bench.py --workload cfg3 --steps 200 --warmup 20 --no-cpu-baseline --kernel-times 2>/tmp/kt.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('label_in_pair $f', d['ms_per_step'], d['value'])"
grep -E "pair_bwd|label_bwd" /tmp/kt.txt
done; done
