#!/bin/bash
# A/B of two builds of the library on the backbone GEMM shapes (run on the GPU box from the repo root):
#   tools/lib_ab.sh <other.so> [act]     -- the in-tree library against <other.so>, alternating, twice each
OTHER=$1; ACT=${2:-0}
for rep in 1 2; do
  echo "== in-tree library (act $ACT)";  GEMM_BENCH_ACT=$ACT python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu.ids | grep -E "^s[0-2]\.|weighted"
  echo "== $OTHER (act $ACT)"; AURORA_HIP_LIB=$OTHER GEMM_BENCH_ACT=$ACT python tools/gemm_bench.py bf16 2>&1 | grep -v amdgpu.ids | grep -E "^s[0-2]\.|weighted"
done
