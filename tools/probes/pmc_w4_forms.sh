set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_pmc_w4_forms.txt
mkdir -p $ROOT/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
export AURORA_HIP_LIB=$ROOT/aurora_amd/_lib/libaurora_hip_w4.so AURORA_GEMM_W4_MIN_K=512
i=0
for st in 4 6; do
 for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf /tmp/pw_$i
  AURORA_GEMM_W4_STAGES=$st timeout 90 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pw_$i -o p -- python $ROOT/tools/gemm_bench.py bf16 sq8192 > /tmp/pw_$i.log 2>&1
  f=$(find /tmp/pw_$i -name "*counter_collection.csv" | head -1)
  echo "== AURORA_GEMM_W4_STAGES=$st (4: LDS-DMA ring, 64-byte pieces; 6: registers, whole 128-byte lines)  pass: $pass" >> $OUT
  if [ -n "$f" ]; then timeout 60 python - "$f" >> $OUT <<'PY'
import collections, csv, sys
disp = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "linear_kernel" not in n: continue
    d = disp.setdefault((r["Dispatch_Id"], n.split("::")[-1][:34]), collections.Counter())
    d[r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (did, k), d in disp.items():
    for c, v in d.items(): agg[k][c].append(v)
for k in sorted(agg):
    print("  %-36s " % k + "  ".join("%s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(agg[k].items())))
PY
  else tail -3 /tmp/pw_$i.log >> $OUT; fi
 done
done
cat $OUT
