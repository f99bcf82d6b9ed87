#!/bin/bash
# Shader clock and socket power while (a) the forecast step runs back to back, (b) one long-K GEMM shape runs back to back
# (rocm-smi sampled once a second beside the load).
cd "$(dirname "$0")/../.." || exit 1
sample() { for i in $(seq 1 ${1:-6}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed -e "s/.*sclk clock level: [^(]*(/sclk /" -e "s/Mhz)//" -e "s/.*Power (W): /W /" | tr "\n" " "; echo; sleep 1; done; }
echo "== idle"; sample 2
echo "== forecast step, back to back"
python bench.py --steps 150 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &
P=$!; sleep 12; sample 30; wait $P
echo "== s2.fc2 (16200 x 2048 x 8192) back to back"
python - <<'PY' &
import sys, torch
sys.path.insert(0, ".")
from aurora_amd.engine import lib
a = (torch.rand(16200, 8192, device="cuda") * 2 - 1).bfloat16(); w = ((torch.rand(2048, 8192, device="cuda") * 2 - 1) / 90).bfloat16()
b = torch.rand(2048, device="cuda"); out = torch.empty(16200, 2048, device="cuda", dtype=torch.bfloat16)
for _ in range(40000): lib.linear(a, w, b, out)
torch.cuda.synchronize()
PY
P=$!; sleep 8; sample; wait $P
