#!/bin/bash
# A/B of the kernel variants selectable at run time (one process per variant: the switches are read once).
#   tools/kernel_ab.sh  ->  gpurun_out/ab_*.log
OUT=gpurun_out; mkdir -p $OUT
SHAPES=s0.qkv,s0.proj,s0.fc1,s0.fc2,s1.qkv,s1.proj,s1.fc1,s1.fc2,s2.qkv,s2.proj,s2.fc1,s2.fc2,sq8192
for v in 0 2 3; do AURORA_ATTN_VARIANT=$v timeout 240 python tools/attn_check.py dump /tmp/attn_v$v.pt; done
timeout 240 python tools/attn_check.py cmp /tmp/attn_v0.pt /tmp/attn_v2.pt > $OUT/ab_attn_check.log 2>&1
timeout 240 python tools/attn_check.py cmp /tmp/attn_v0.pt /tmp/attn_v3.pt >> $OUT/ab_attn_check.log 2>&1
for v in 0 2 3; do echo "== AURORA_ATTN_VARIANT=$v"; AURORA_ATTN_VARIANT=$v timeout 240 python tools/attn_bench.py; done > $OUT/ab_attn_bench.log 2>&1
for v in 1 3; do echo "== AURORA_GEMM_VARIANT=$v (no activation)"; AURORA_GEMM_VARIANT=$v timeout 240 python tools/gemm_bench.py bf16 $SHAPES; done > $OUT/ab_gemm.log 2>&1
for v in 1 3; do echo "== AURORA_GEMM_VARIANT=$v (GELU epilogue, fc1 shapes)"; AURORA_GEMM_VARIANT=$v GEMM_BENCH_ACT=1 timeout 240 python tools/gemm_bench.py bf16 s0.fc1,s1.fc1,s2.fc1; done >> $OUT/ab_gemm.log 2>&1
echo "== fp32 decoder shapes" >> $OUT/ab_gemm.log; GEMM_BENCH_ACT=1 timeout 240 python tools/gemm_bench.py f32 d.fc1,d.proj >> $OUT/ab_gemm.log 2>&1
tail -n 60 $OUT/ab_attn_check.log $OUT/ab_attn_bench.log $OUT/ab_gemm.log
