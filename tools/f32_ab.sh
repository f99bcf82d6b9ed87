#!/bin/bash
# A/B of the two-fp16-term fp32 GEMM kernels on the encoder/decoder shapes of the 0.25-degree step.  Run on the GPU box
# from the repo root.  AURORA_F32_VARIANT=0: in-phase 256 x 256 ring, 1: ping-pong 128 x 256; GEMM_BENCH_PRESPLIT: which of
# weights (w), activations (a), result (c) are in the fp16-pair layout (ping-pong kernel only).
for v in 0 1; do
  echo "== AURORA_F32_VARIANT=$v, split in the kernel"
  AURORA_F32_GEMM=2 AURORA_F32_VARIANT=$v python tools/gemm_bench.py f32 f32shapes
done
for pre in w aw awc; do
  echo "== ping-pong kernel, pre-split: $pre"
  AURORA_F32_GEMM=2 GEMM_BENCH_PRESPLIT=$pre python tools/gemm_bench.py f32 f32shapes
done
