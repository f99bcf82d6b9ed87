"""Where does perceiver_out differ from attention + to_out?  Error pattern by feature, column, level."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from aurora_amd.engine import lib  # noqa: E402

cols, Lq, Lk, heads, hd, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 13, 3, 16, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
inner = heads * hd
g = torch.Generator().manual_seed(1)
q = torch.rand(Lq, inner, generator=g, dtype=torch.float64) * 2 - 1
kv = torch.rand(Lk * cols, 2 * inner, generator=g, dtype=torch.float64) * 2 - 1
w = (torch.rand(N, inner, generator=g, dtype=torch.float64) * 2 - 1) * inner ** -0.5
kvr = kv.reshape(1, Lk, cols, 2, heads, hd).permute(3, 0, 2, 4, 1, 5)
qq = q.reshape(Lq, heads, hd).permute(1, 0, 2)[None, None].expand(1, cols, -1, -1, -1)
att = F.scaled_dot_product_attention(qq, kvr[0], kvr[1]).permute(0, 1, 3, 2, 4).reshape(cols * Lq, inner)
ref = F.linear(att, w)
P, Vp = lib.perceiver_probs(q.float().cuda(), kv.float().cuda(), 1, cols, Lk * cols, cols, Lq, Lk, heads, hd)
wp = lib.split_f16(w.float().cuda(), scale=64.0)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    out = torch.full((cols * Lq, N), float("nan"), device="cuda")
    lib.perceiver_out(Vp, wp, P, out, cols, Lq, Lk, heads, hd)
    torch.cuda.synchronize()
    o = out.cpu().double().reshape(cols, Lq, N)
    r = ref.reshape(cols, Lq, N)
    bad = ~((o - r).abs() < 1e-4)
    print(f"rep {rep}: bad {int(bad.sum())} of {bad.numel()}; nan {int(torch.isnan(o).sum())}")
    if bad.any():
        print(" bad per n%128 block of 16:", [int(bad[:, :, :].reshape(cols, Lq, N // 128, 8, 16)[:, :, :, b].sum()) for b in range(8)])
        print(" bad per n%4:", [int(bad[:, :, k::4].sum()) for k in range(4)])
        print(" bad per n-tile of 128:", [int(bad[:, :, t * 128:(t + 1) * 128].sum()) for t in range(N // 128)])
        print(" bad per level:", [int(bad[:, l].sum()) for l in range(Lq)])
        print(" bad per col%32:", [int(bad[c::32].sum()) for c in range(min(32, cols))])
        idx = bad.nonzero()[:8]
        for c, l, n in idx.tolist():
            print(f"   col {c} l {l} n {n}: got {o[c, l, n]:.6f} ref {r[c, l, n]:.6f}")
