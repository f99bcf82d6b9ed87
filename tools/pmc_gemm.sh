#!/bin/bash
# PMC passes over tools/gemm_bench.py (run ON the GPU box):  tools/pmc_gemm.sh <dtype> <shapes> <tag>
# Writes gpurun_out/pmc_gemm_<tag>.json (per-kernel, per-launch means).
set -u
DT=$1; SHAPES=$2; TAG=$3
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
ARGS=()
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVES" \
            "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$i -o p -- python $ROOT/tools/gemm_bench.py $DT $SHAPES > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && ARGS+=("p$i=$f")
  grep -v simple_timer /tmp/pmc_$i.log | tail -2
done
python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_gemm_$TAG.json "${ARGS[@]}" > /dev/null
python - <<PY
import json
d = json.load(open("$ROOT/gpurun_out/pmc_gemm_$TAG.json"))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        if c != "launches": print(f"   {c:40s} {x:16.1f}")
PY
