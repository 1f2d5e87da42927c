#!/bin/bash
# A/B inside one GPU session: the round-2 tree under abtest/r02 against the working tree, alternating, N rounds.
N=${1:-3}; W=${2:-cfg3}
for i in $(seq $N); do
  (cd abtest/r02 && python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --kernel-times 2>/tmp/kt_old.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r02 ', d['ms_per_step'], d['value'])"; grep -E "lstm_pair_(fwd|bwd)" /tmp/kt_old.txt)
  python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --kernel-times 2>/tmp/kt_new.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['value'])"; grep -E "lstm_pair_(fwd|bwd)" /tmp/kt_new.txt
done
