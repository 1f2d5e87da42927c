"""Aggregate rocprofv3 --pmc SQ passes (counter_collection CSVs) into per-kernel busy fractions.
Usage: python tools/pmc_sq.py out.json pass1.csv [pass2.csv ...]
Per kernel (average over its dispatches): every counter's sum, and where the inputs are there
  valu_issue      = 2 * SQ_INSTS_VALU / (4 * SQ_BUSY_CU_CYCLES)   (a wave64 VALU instruction holds a SIMD32 for 2 cycles; 4 SIMDs per
                    CU; SQ_BUSY_CU_CYCLES = cycles summed over the CUs that had waves: 98 us x 2.1 GHz x 256 CUs for pair_fwd)
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4   (per SIMD: 4 matrix pipes per CU)
  lds_conflict    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (or / SQ_ACTIVE_INST_LDS when IDX_ACTIVE was not collected)
Units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles."""
import collections, csv, json, re, sys

out, paths = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in paths:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = {}
    for r in csv.DictReader(open(path)):
        per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name']
    for d, cs in per.items():
        k = re.sub(r'^void ', '', name[d]).replace('clv::', '')
        k = re.sub(r'\(.*', '', k)[:100]
        for c, v in cs.items():
            acc[k][c].append(v)
rows = []
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    row = dict(kernel=k, dispatches=max(len(v) for v in cs.values()), counters={c: round(v, 1) for c, v in sorted(m.items())})
    busy = m.get('SQ_BUSY_CU_CYCLES') or m.get('SQ_BUSY_CYCLES')
    if busy:
        if 'SQ_INSTS_VALU' in m:
            row['valu_issue'] = round(2 * m['SQ_INSTS_VALU'] / (4 * busy), 4)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            row['mfma_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / busy / 4, 4)
    if 'SQ_LDS_BANK_CONFLICT' in m:
        den = m.get('SQ_LDS_IDX_ACTIVE') or m.get('SQ_ACTIVE_INST_LDS')
        if den:
            row['lds_conflict_per_lds_cycle'] = round(m['SQ_LDS_BANK_CONFLICT'] / den, 4)
    if 'SQ_WAVE_CYCLES' in m and 'SQ_WAIT_ANY' in m:
        row['wait_any_frac'] = round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4)
    rows.append(row)
rows.sort(key=lambda r: -r['counters'].get('SQ_BUSY_CU_CYCLES', r['counters'].get('SQ_BUSY_CYCLES', r['counters'].get('SQ_WAVE_CYCLES', 0))))
json.dump(dict(note=__doc__.split('\n')[0] + " Fractions are per-kernel averages over dispatches of bench.py --no-graph.",
               passes=[p.split('/')[-2] for p in paths], kernels=rows), open(out, 'w'), indent=1)
for r in rows[:8]:
    print("%-60s valu_issue %-7s mfma_busy %-7s lds_conflict %-7s" % (r['kernel'][:60], r.get('valu_issue'), r.get('mfma_busy'),
                                                                  r.get('lds_conflict_per_lds_cycle')))
