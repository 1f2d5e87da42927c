"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per kernel launch.
gfx950 corrections per MI355X_MICROARCH.md: both counters are in KB (x1024); FETCH_SIZE reports half of the bytes read (x2)."""
import collections, csv, json, re, sys

workload, fetch_csv, write_csv, out = sys.argv[1:5]


def per_kernel(path, counter):
    disp = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        disp[r['Dispatch_Id']] += float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name']
    agg = collections.defaultdict(list)
    for d, v in disp.items():
        agg[name[d]].append(v)
    return agg


f, w = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
rows = []
for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) * 2 + sum(w.get(k, [0])))):
    fk, wk = f.get(k, [0.0]), w.get(k, [0.0])
    short = re.sub(r'^void ', '', k).replace('clv::', '')
    rows.append(dict(kernel=short[:120], launches=len(fk), fetch_size_kb_avg=sum(fk) / len(fk), write_size_kb_avg=sum(wk) / len(wk),
                     hbm_read_bytes_corrected=2 * 1024 * sum(fk) / len(fk), hbm_write_bytes=1024 * sum(wk) / len(wk)))
dom = [r for r in rows if re.match(r'lstm_(pair_|mx_)?(fwd|bwd)(_mfma)?_kernel', r['kernel'])]      # the kernels bench.py's roofline times
per_launch = sum(r['hbm_read_bytes_corrected'] + r['hbm_write_bytes'] for r in dom) / max(len(dom), 1)
# steps the profiled run executed (set-up + warm-up + timed): the launch count of a kernel that runs once per LSTM pass
# (bench.py's set-up steps changed in round 3; a number on the command line is only the fallback)
per_step = {'lstm_pair_fwd_kernel': 1, 'lstm_mx_fwd_kernel': 2, 'lstm_fwd_mfma_kernel': 2, 'vae_fused_kernel': 1}
steps = None
for r in rows:
    for key, n in per_step.items():
        if steps is None and r['kernel'].startswith(key):
            same = sum(x['launches'] for x in rows if x['kernel'].startswith(key))
            steps = same // n
if not steps:
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 9
step_bytes = sum((r['hbm_read_bytes_corrected'] + r['hbm_write_bytes']) * r['launches'] for r in rows) / steps
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --no-graph); FETCH_SIZE x2 "
                    "(gfx950 reports half of the bytes read), both KB -> bytes x1024 (MI355X_MICROARCH.md)",
               workload=workload, dominant_kernels=[r['kernel'] for r in dom], dominant_bytes_per_launch=per_launch,
               step_bytes=step_bytes, steps_profiled=steps,
               kernels=rows), open(out, 'w'), indent=1)
print("HBM bytes per step %.0f" % step_bytes)
print("dominant kernels:", [r['kernel'][:40] for r in dom], "bytes/launch %.0f" % per_launch)
