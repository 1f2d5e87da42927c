"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per kernel launch: bench.aggregate_pmc (the
gfx950 corrections of MI355X_MICROARCH.md) + the sha of the kernel sources the passes ran on, so that bench.py can tell a
stale summary.  Usage: python tools/pmc_traffic.py <workload> <fetch.csv> <write.csv> <out.json> [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

workload, fetch_csv, write_csv, out = sys.argv[1:5]
d = bench.aggregate_pmc(workload, fetch_csv, write_csv, steps_fallback=int(sys.argv[5]) if len(sys.argv) > 5 else 9)
d['sources_sha16'] = bench.sources_sha16()
json.dump(d, open(out, 'w'), indent=1)
print("HBM bytes per step %.0f" % d['step_bytes'])
print("dominant kernels:", [k[:40] for k in d['dominant_kernels']], "bytes/launch %.0f" % d['dominant_bytes_per_launch'])
