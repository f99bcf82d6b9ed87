"""The decoder's level de-aggregation at 0.25 degree (64,800 columns, 13 levels, 3 latent keys, 16 x 64, D = 1024):
the re-associated pair (perceiver_probs + perceiver_out) against attention + to_out on pre-split operands."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aurora_amd.engine import lib  # noqa: E402

COLS, LQ, LK, HEADS, HD, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64800, 13, 3, 16, 64, 1024
INNER = HEADS * HD


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


g = torch.Generator(device="cuda").manual_seed(1)
q = torch.rand(LQ, INNER, device="cuda", generator=g) * 2 - 1
kv = torch.rand(LK * COLS, 2 * INNER, device="cuda", generator=g) * 2 - 1
w = (torch.rand(N, INNER, device="cuda", generator=g) * 2 - 1) * INNER ** -0.5
w_pairs = lib.split_f16(w, scale=64.0)
att = torch.empty(COLS * LQ, INNER, device="cuda")
out_a = torch.empty(COLS * LQ, N, device="cuda")
out_b = torch.empty(COLS * LQ, N, device="cuda")
word = torch.zeros(1, device="cuda")
P, Vp = lib.perceiver_probs(q, kv, 1, COLS, LK * COLS, COLS, LQ, LK, HEADS, HD)


def plain_attention():
    lib.perceiver_attention(q, 0, kv, att, 1, COLS, LK * COLS, COLS, LQ, LK, HEADS, HD, pair_guard=(word, 1.0))


def plain_to_out():
    lib.linear(att, w_pairs, None, out_a, presplit=lib.F32_A_SPLIT | lib.F32_W_SPLIT)


def probs():
    lib.load().aurora_hip_perceiver_probs(q.data_ptr(), kv.data_ptr(), P.data_ptr(), Vp.data_ptr(), 1, COLS, LK * COLS, COLS, LQ, LK,
                                          HEADS, HD, None, 0.0, None)


def out():
    lib.perceiver_out(Vp, w_pairs, P, out_b, COLS, LQ, LK, HEADS, HD)


plain_attention(); plain_to_out(); probs(); out()
torch.cuda.synchronize()
err = ((out_a - out_b).abs().max() / out_a.abs().max()).item()
t = {k: timed(f) for k, f in (("attention", plain_attention), ("to_out", plain_to_out), ("probs", probs), ("out", out))}
flop = 2.0 * COLS * LK * N * INNER + 2.0 * COLS * LQ * N * HEADS * LK
print(f"cols={COLS}: attention {t['attention']:.3f} + to_out {t['to_out']:.3f} = {t['attention'] + t['to_out']:.3f} ms | "
      f"probs {t['probs']:.3f} + out {t['out']:.3f} = {t['probs'] + t['out']:.3f} ms "
      f"(out: {flop / t['out'] * 1e-9:.1f} TFLOP/s of its own work, {2.0 * COLS * LQ * N * INNER / t['out'] * 1e-9:.1f} of the GEMM it replaces) | "
      f"max |a - b| / max |a| = {err:.2e}", flush=True)
