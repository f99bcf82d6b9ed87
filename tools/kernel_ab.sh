#!/bin/bash
# A/B of the GEMM kernel variants selectable at run time (one process per variant: the switches are read once).
#   tools/kernel_ab.sh  ->  gpurun_out/ab_gemm.log
OUT=gpurun_out; mkdir -p $OUT
SHAPES=s0.qkv,s0.proj,s0.fc1,s0.fc2,s1.qkv,s1.proj,s1.fc1,s1.fc2,s2.qkv,s2.proj,s2.fc1,s2.fc2,sq4096,sq8192
for v in 4 5 7; do echo "== AURORA_GEMM_VARIANT=$v (1 ring+prio, 4 default: persistent by shape rule, 5 ping-pong, 7 ping-pong + persistent)"; AURORA_GEMM_VARIANT=$v timeout 240 python tools/gemm_bench.py bf16 $SHAPES; done > $OUT/ab_gemm.log 2>&1
grep -v amdgpu $OUT/ab_gemm.log
