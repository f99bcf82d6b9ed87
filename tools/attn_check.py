"""Dump / compare the bf16 window-attention output of the current AURORA_ATTN_VARIANT on fixed random inputs:
    python tools/attn_check.py dump /tmp/attn_v0.pt        (under each variant)
    python tools/attn_check.py cmp /tmp/attn_v0.pt /tmp/attn_v2.pt
Stages 0-2 of the 0.25-degree grid (stage 2 has zero-padded windows), shifted and not, plus a clamped-window case."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def dump(path):
    from aurora_amd.engine import geometry, lib

    outs = {}
    cases = [("s0", (4, 180, 360), 512, 8, (2, 6, 12)), ("s1", (4, 90, 180), 1024, 16, (2, 6, 12)),
             ("s2", (4, 45, 90), 2048, 32, (2, 6, 12)), ("small", (4, 4, 8), 256, 4, (2, 4, 8)),
             ("odd", (4, 13, 26), 128, 2, (2, 6, 12))]
    for name, res, D, heads, ws in cases:
        L = res[0] * res[1] * res[2]
        g = torch.Generator(device="cuda").manual_seed(len(name) + D)
        qkv = torch.randn(L, 3 * D, device="cuda", generator=g).bfloat16()
        bias = torch.randn(3 * D, device="cuda", generator=g)
        for shifted in (False, True):
            tok, grp, _ = geometry.window_tables(res, ws, shifted)
            tok_d = torch.from_numpy(tok).cuda()
            grp_d = None if grp is None else torch.from_numpy(grp).cuda()
            out = torch.full((L, D), float("nan"), device="cuda", dtype=torch.bfloat16)
            lib.window_attention(qkv, bias, out, tok_d, grp_d, 1, L, D, heads)
            torch.cuda.synchronize()
            outs[f"{name}.{int(shifted)}"] = out.cpu()
    torch.save(outs, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in A:
        fa, fb = A[k].float(), B[k].float()
        same = torch.equal(A[k].view(torch.int16), B[k].view(torch.int16))
        d = (fa - fb).abs().max().item()
        print(f"{k:10s} identical={same} max|diff|={d:.3e} finite={bool(torch.isfinite(fb).all())}")
        bad += (not same)
    print("ATTN_CHECK", "OK" if not bad else f"{bad} cases differ")


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
