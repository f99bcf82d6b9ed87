#!/bin/bash
# In-step A/B of environment switches (run on the GPU box from the repo root):
#   tools/env_ab.sh "A=1" "A=0" "A=2 B=64" ...   -- bench.py (10 steps) under each setting, the whole list twice
for rep in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); k = d['kernel_ms_per_step']
print('%-40s %7.2f ms/step | bf16 gemm %6.2f  f32 gemm %6.2f  layernorm %6.2f  gemm+ln %5.2f  attention %5.2f  perceiver %5.2f' % (sys.argv[1], d['ms_per_step'], k['linear_bf16'], k['linear_f32'], k['layernorm'], k.get('linear_layernorm_bf16', 0.0), k['window_attention_bf16'], k['perceiver_attention']))" "$setting"
  done
done
