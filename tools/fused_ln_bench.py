"""Stage-0 linear + AdaLN pairs (M = 259,200, D = 512): one fused launch against linear followed by layernorm."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aurora_amd.engine import lib  # noqa: E402

M, N = 259200, 512


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, K in (("proj", 512), ("fc2", 2048)):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).bfloat16()
    b, gain, shift = (torch.rand(N, device="cuda") for _ in range(3))
    x = torch.randn(M, N, device="cuda")
    xb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(xb)
    big = torch.empty(1 << 28, device="cuda", dtype=torch.float32)   # 1 GiB: evicts the 256 MB Infinity Cache between runs

    def separate():
        lib.linear(a, w, b, y)
        lib.layernorm(y, gain, shift, res=x, out_f32=x, out_t=xb)

    def fused():
        lib.linear_layernorm(a, w, b, gain, shift, x, x, xb)

    def lin():
        lib.linear(a, w, b, y)

    def ln():
        lib.layernorm(y, gain, shift, res=x, out_f32=x, out_t=xb)

    print(f"{name} K={K}: linear {timed(lin):.3f} ms + layernorm {timed(ln):.3f} ms = separate {timed(separate):.3f} ms | fused {timed(fused):.3f} ms",
          flush=True)
