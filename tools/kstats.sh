#!/bin/bash
# rocprofv3 kernel averages of one bench run, per step: bash tools/kstats.sh <tag> [bench args]
TAG=${1:-k}; shift
export TMPDIR=/tmp; R=$PWD; G=$R/gpurun_out/$TAG; mkdir -p $G
(cd /tmp && rocprofv3 --kernel-trace --stats -d $G/prof -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 200 --warmup 20 "$@" > $G/prof.log 2>&1)
tail -1 $G/prof.log | cut -c1-160
cp $G/prof/p_kernel_stats.csv $G/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$G/kernel_stats.csv")))
# profiled steps = the launch count of a once-per-step kernel (bench.py also runs set-up steps; "$@" may change --steps)
once=[int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ("lstm_pair_fwd", "vae_fused_kernel", "vrnn_label_fwd", "out_head_"))]
steps=min(once) if once else 223
print("profiled steps: %d" % steps)
tot=0
for r in rows:
    if "clv::" in r["Name"]:
        per=float(r["TotalDurationNs"])/steps/1e3
        tot+=per
        print("%-62s %5s avg %7.2f us  per-step %7.2f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, per))
print("sum per step %.1f us" % tot)
PY
