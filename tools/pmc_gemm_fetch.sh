#!/bin/bash
# Which operand is over-fetched?  FETCH_SIZE (L2 -> fabric reads, Infinity-Cache hits included) of the bf16 GEMM, ONE shape
# per rocprofv3 run (run ON the GPU box):   tools/pmc_gemm_fetch.sh <tag> shape [shape ...]
# Prints per shape: bytes read per launch (2 x FETCH_SIZE, profiles/r02_fetch_calibration.txt) against the bytes of A
# (M x K) and of W (N x K) -- reads ~ A + W: none; ~ 2 A: the activation lines are fetched twice (half lines per K-stage);
# ~ A + g W: the weight panel is re-fetched per m-tile group.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_gemm_fetch_$TAG.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for SH in "$@"; do
  rm -rf /tmp/pf
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o p -- python $ROOT/tools/gemm_bench.py bf16 $SH > /tmp/pf.log 2>&1
  f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
  python - "$SH" "$f" "$ROOT" >> $OUT <<'PY'
import csv, sys
sys.path.insert(0, sys.argv[3] + "/tools")
shape, path = sys.argv[1], sys.argv[2]
M = N = K = None
for ln in open(sys.argv[3] + "/tools/gemm_bench.py"):
    if f'("{shape}",' in ln:
        part = ln[ln.index(f'("{shape}",'):].split(")")[0].split(",")
        M, N, K = int(part[1]), int(part[2]), int(part[3])
disp = {}
for r in csv.DictReader(open(path)):
    if "linear_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        disp[r["Dispatch_Id"]] = disp.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
vals = list(disp.values())
read = 2.0 * 1024 * sum(vals) / max(len(vals), 1)
A, W = M * K * 2, N * K * 2
print(f"{shape:8s} M={M:6d} N={N:5d} K={K:5d}  launches {len(vals)}  read {read / 1e6:8.1f} MB   A {A / 1e6:7.1f} MB  W {W / 1e6:6.1f} MB   read/(A+W) {read / (A + W):5.2f}   (read-A)/W {(read - A) / W:7.1f}")
PY
done
cat $OUT
