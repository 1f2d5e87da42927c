export CLV_LSTM_MX=1 TMPDIR=/tmp
R=$PWD; G=$R/gpurun_out/r04_mx_lds; mkdir -p $G
(cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $G -o s --output-format csv -- python3 $R/tools/mx_bench.py 1024 256 32 > $G/log.txt 2>&1)
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$G/s_counter_collection.csv")):
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    if 'lstm' not in k: continue
    m = {n: sum(v)/len(v) for n, v in c.items()}
    print("%-60s conflict/active %.3f  lds_active/busy_cu %.3f  insts_lds %.0f" % (k, m['SQ_LDS_BANK_CONFLICT']/max(m['SQ_LDS_IDX_ACTIVE'],1), m['SQ_LDS_IDX_ACTIVE']/max(m['SQ_BUSY_CU_CYCLES'],1), m['SQ_INSTS_LDS']))
PY
