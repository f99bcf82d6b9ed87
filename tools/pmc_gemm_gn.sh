#!/bin/bash
# L2 -> fabric reads (FETCH_SIZE) and time of one bf16 GEMM shape under different n-group widths of the tile order
# (AURORA_GEMM_GN: an XCD's round of 32 tiles is (32 / gn) m-tiles x gn n-tiles), no Python:  tools/pmc_gemm_gn.sh <tag>
set -u
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_pmc_gemm_gn.txt
echo "# shape MxNxK | gn | read MB per launch (2 x FETCH_SIZE x 1024 B) | read / (A + W) | us per launch (un-profiled run)" > $OUT
cd /tmp && export TMPDIR=/tmp
for SH in 64800,4096,1024 64800,3072,1024 64800,1024,4096 16200,8192,2048 16200,6144,2048 16200,2048,8192 259200,512,2048; do
  for GN in 8 4 2; do
    rm -rf /tmp/pf
    AURORA_GEMM_PP=0 AURORA_GEMM_GN=$GN AURORA_GEMM_GN_K=0 CHECK_NO_WS=1 CHECK_REPS=5 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- $ROOT/tools/probes/gemm_check shape=$SH > /tmp/pf.log 2>&1
    US=$(AURORA_GEMM_PP=0 AURORA_GEMM_GN=$GN AURORA_GEMM_GN_K=0 CHECK_NO_WS=1 $ROOT/tools/probes/gemm_check shape=$SH | grep " us " | awk '{print $6}')
    f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
    python3 - "$SH" "$GN" "$f" "$US" >> $OUT <<'PY'
import csv, sys
M, N, K = (int(x) for x in sys.argv[1].split(","))
disp = {}
for r in csv.DictReader(open(sys.argv[3])):
    if "linear_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        disp[r["Dispatch_Id"]] = disp.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
vals = list(disp.values())
read = 2.0 * 1024 * sum(vals) / max(len(vals), 1)
print(f"{M}x{N}x{K}  gn {sys.argv[2]}  read {read / 1e6:8.1f} MB  x{read / ((M + N) * K * 2):5.2f}  {sys.argv[4]} us  ({len(vals)} launches)")
PY
  done
done
cat $OUT
