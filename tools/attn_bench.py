"""Throughput of the bf16 window-attention kernel at the three backbone stages of the 0.25-degree config."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aurora_amd.engine import geometry, lib  # noqa: E402

STAGES = [("s0", (4, 180, 360), 512, 8, 12), ("s1", (4, 90, 180), 1024, 16, 20), ("s2", (4, 45, 90), 2048, 32, 16)]
which = sys.argv[1].split(",") if len(sys.argv) > 1 else None
tot = 0.0
for name, res, D, heads, wt in STAGES:
    if which and name not in which:
        continue
    L = res[0] * res[1] * res[2]
    qkv = (torch.randn(L, 3 * D, device="cuda")).bfloat16()
    bias = torch.randn(3 * D, device="cuda")
    out = torch.empty(L, D, device="cuda", dtype=torch.bfloat16)
    for shifted in (False, True):
        tok, grp, _ = geometry.window_tables(res, (2, 6, 12), shifted)
        tok_d = torch.from_numpy(tok).cuda()
        grp_d = None if grp is None else torch.from_numpy(grp).cuda()
        for _ in range(2):
            lib.window_attention(qkv, bias, out, tok_d, grp_d, 1, L, D, heads)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            lib.window_attention(qkv, bias, out, tok_d, grp_d, 1, L, D, heads)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = 4.0 * tok.shape[0] * tok.shape[1] * D * 2
        tot += ms * wt / 2
        print(f"{name} shifted={int(shifted)} windows={tok.shape[0]:5d} {ms:7.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
print(f"weighted per step: {tot:.2f} ms")
