#!/bin/bash
# L2 / fabric counters of OUR bf16 GEMM and the vendor library's on the same shapes (run ON the GPU box):
#   tools/probes/pmc_vendor_compare.sh "sq8192,s1.fc2"      -> gpurun_out/r05_pmc_vendor_compare.txt
# Two separate --pmc passes (never combined with the trace domains gpurun refuses).  A YARDSTICK: nothing in aurora_amd/ calls the
# vendor GEMM.
set -u
SHAPES=${1:-sq8192,s1.fc2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_pmc_vendor_compare.txt
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
i=0
for pass in "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pv_$i
  GEMM_BENCH_VENDOR=1 timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pv_$i -o p -- python $ROOT/tools/gemm_bench.py bf16 $SHAPES > /tmp/pv_$i.log 2>&1
  f=$(find /tmp/pv_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $pass  ($(grep -c . "$f" 2>/dev/null) rows)" >> $OUT
  if [ -n "$f" ]; then
    timeout 60 python - "$f" >> $OUT <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "linear_kernel" not in n and "Cijk" not in n:
        continue
    short = ("ours   " + n.split("::")[-1][:28]) if "linear_kernel" in n else ("vendor " + n[n.find("MT"):n.find("MT") + 14])
    key = (short, r.get("Grid_Size", "?"))
    d = disp.setdefault((r["Dispatch_Id"], key), collections.Counter())
    d[r["Counter_Name"]] += float(r["Counter_Value"])
for (did, key), d in disp.items():
    for c, v in d.items():
        agg[key][c].append(v)
for key in sorted(agg):
    print("  %-46s grid %-10s " % key + "  ".join("%s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(agg[key].items())))
PY
  else
    tail -3 /tmp/pv_$i.log >> $OUT
  fi
done
cat $OUT
