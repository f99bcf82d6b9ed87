"""Which weight row does output feature n see?  W[n, k] = (n + 1) at k = K0 only -> out[., n] / att[., K0] = n + 1."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from aurora_amd.engine import lib  # noqa: E402

cols, Lq, Lk, heads, hd, N = 32, 13, 3, 16, 64, 128
inner = heads * hd
g = torch.Generator().manual_seed(1)
q = torch.rand(Lq, inner, generator=g, dtype=torch.float64) * 2 - 1
kv = torch.rand(Lk * cols, 2 * inner, generator=g, dtype=torch.float64) * 2 - 1
kvr = kv.reshape(1, Lk, cols, 2, heads, hd).permute(3, 0, 2, 4, 1, 5)
qq = q.reshape(Lq, heads, hd).permute(1, 0, 2)[None, None].expand(1, cols, -1, -1, -1)
att = F.scaled_dot_product_attention(qq, kvr[0], kvr[1]).permute(0, 1, 3, 2, 4).reshape(cols * Lq, inner)
P, Vp = lib.perceiver_probs(q.float().cuda(), kv.float().cuda(), 1, cols, Lk * cols, cols, Lq, Lk, heads, hd)
for K0 in (0, 5, 40, 64 * 3 + 9, 1023):
    w = torch.zeros(N, inner, dtype=torch.float64)
    w[:, K0] = torch.arange(1, N + 1, dtype=torch.float64) / 64
    wp = lib.split_f16(w.float().cuda(), scale=64.0)
    out = torch.full((cols * Lq, N), float("nan"), device="cuda")
    lib.perceiver_out(Vp, wp, P, out, cols, Lq, Lk, heads, hd)
    torch.cuda.synchronize()
    ratio = (out.cpu().double() / att[:, K0:K0 + 1] * 64)
    r0 = ratio[0]
    print(f"K0={K0}: row 0 ratios:", [round(float(x), 2) for x in r0[[0, 1, 2, 3, 62, 63, 64, 65, 66, 67, 68, 80, 96, 112, 126, 127]]])
    print("   consistent over rows:", bool(((ratio - r0).abs() < 1e-2).all()), " n>=64 ratios (first 16):", [round(float(x), 1) for x in r0[64:80]])
