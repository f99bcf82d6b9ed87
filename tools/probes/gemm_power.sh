#!/bin/bash
# clock, power and rate of ONE bf16 GEMM shape run back to back for ~8 s: random operands against all-zero operands
# (GEMM_POWER_SHAPE="M N K", default 8192 8192 8192)
cd "$(dirname "$0")/../.." || exit 1
sample() { for i in $(seq 1 ${1:-4}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed -e "s/.*sclk clock level: [^(]*(/sclk /" -e "s/Mhz)//" -e "s/.*Power (W): /W /" | tr "\n" " "; echo; sleep 1; done; }
run() {   # $1 = 0 / 1: random / zero operands; the sampler runs beside it, samples under load (> 500 W) are printed
  : > /tmp/gemm_power_samples.txt
  ( while true; do sample 1 >> /tmp/gemm_power_samples.txt; done ) & S=$!
  ZERO=$1 python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, ".")
from aurora_amd.engine import lib
M, N, K = (int(x) for x in os.environ.get("GEMM_POWER_SHAPE", "8192 8192 8192").split())
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).bfloat16()
if os.environ["ZERO"] == "1":
    a.zero_(); w.zero_()
b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(20): lib.linear(a, w, b, out)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 8.0:
    for _ in range(200): lib.linear(a, w, b, out)
    torch.cuda.synchronize(); n += 200
dt = time.perf_counter() - t0
print(f"operands {'zero' if os.environ['ZERO'] == '1' else 'random'}: {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s sustained over {dt:.1f} s", flush=True)
PY
  kill $S 2>/dev/null; wait $S 2>/dev/null
  awk '$4 > 500' /tmp/gemm_power_samples.txt | tail -4
}
for zero in 0 1 0 1; do run $zero; done
