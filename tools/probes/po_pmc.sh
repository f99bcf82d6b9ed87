#!/bin/bash
# cache / LDS / issue counters of perceiver_out_kernel on the 0.25-degree shape (tools/perceiver_out_bench.py)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/po_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/po_pmc -o p -- python $ROOT/tools/perceiver_out_bench.py > /tmp/po.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/po_pmc/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'perceiver_out_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'per launch', sum(v) / len(v), 'n', len(v))
PY
done
