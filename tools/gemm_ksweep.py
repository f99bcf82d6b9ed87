"""Time of aurora_hip_linear vs K at fixed (M, N): separates the per-tile fixed cost from the per-stage cost."""
import sys
from pathlib import Path

import os

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aurora_amd.engine import lib  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
M, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (259200, 2048)
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).to(dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    for _ in range(2):
        lib.linear(a, w, None, out, act=int(os.environ.get("KSWEEP_ACT", "0")))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.linear(a, w, None, out, act=int(os.environ.get("KSWEEP_ACT", "0")))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tiles = -(-M // 256) * -(-N // 256)
    print(f"K={K:5d} {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TF/s   per 256x256 tile-round: "
          f"{ms * 1e3 / (tiles / 256):7.2f} us", flush=True)
    del a, w, out
