#!/bin/bash
# A/B of the bf16 GEMM variants on one GPU box (no Python):  tools/gpu_gemm_ab.sh  -> gpurun_out/gemm_ab.log
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/gemm_ab.log
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 180 tools/probes/gemm_check $SETS >> $out 2>&1; echo "rc=$?" >> $out; }
SETS="small step" run AURORA_GEMM_PP=0 CHECK_NO_WS=1
SETS="small step band8" run AURORA_GEMM_PP=2 CHECK_NO_WS=1
SETS="step" run AURORA_GEMM_PP=6 CHECK_NO_WS=1
SETS="step" run AURORA_GEMM_PP=0 CHECK_NO_WS=1
SETS="step" run AURORA_GEMM_PP=2 CHECK_NO_WS=1
SETS="band8 band4 band2" run AURORA_GEMM_PP=0
tail -3 $out
