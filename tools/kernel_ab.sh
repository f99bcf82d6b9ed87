#!/bin/bash
# A/B of the kernel variants selectable at run time (one process per variant: the switches are read once).
#   tools/kernel_ab.sh  ->  gpurun_out/ab_*.log
OUT=gpurun_out; mkdir -p $OUT
for v in 0 1 2; do AURORA_ATTN_VARIANT=$v timeout 240 python tools/attn_check.py dump /tmp/attn_v$v.pt; done
timeout 240 python tools/attn_check.py cmp /tmp/attn_v0.pt /tmp/attn_v1.pt > $OUT/ab_attn_check.log 2>&1
timeout 240 python tools/attn_check.py cmp /tmp/attn_v0.pt /tmp/attn_v2.pt >> $OUT/ab_attn_check.log 2>&1
for v in 0 1 2; do echo "== AURORA_ATTN_VARIANT=$v"; AURORA_ATTN_VARIANT=$v timeout 240 python tools/attn_bench.py; done > $OUT/ab_attn_bench.log 2>&1
for v in 0 1; do echo "== AURORA_GEMM_VARIANT=$v (no activation)"; AURORA_GEMM_VARIANT=$v timeout 240 python tools/gemm_bench.py bf16 s0.qkv,s0.proj,s0.fc1,s0.fc2,s1.qkv,s1.proj,s1.fc1,s1.fc2,s2.qkv,s2.proj,s2.fc1,s2.fc2,sq8192; done > $OUT/ab_gemm.log 2>&1
for g in 0 1; do echo "== GELU variant $g (fc1 shapes, GELU epilogue)"; AURORA_GELU_VARIANT=$g GEMM_BENCH_ACT=1 timeout 240 python tools/gemm_bench.py bf16 s0.fc1,s1.fc1,s2.fc1; done >> $OUT/ab_gemm.log 2>&1
tail -n 40 $OUT/ab_attn_check.log $OUT/ab_attn_bench.log $OUT/ab_gemm.log
