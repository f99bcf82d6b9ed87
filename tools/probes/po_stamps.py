"""Slot stamps of perceiver_out_kernel (a probe build: AURORA_BUILD_FLAGS=-DPO_STAMPS python -m aurora_amd.build --force): per wave, per K-stage: start of R, end of R, start of C, end of C."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from aurora_amd.engine import lib  # noqa: E402

COLS, LQ, LK, HEADS, HD, N = 64800, 13, 3, 16, 64, 1024
INNER = HEADS * HD
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.rand(LQ, INNER, device="cuda", generator=g) * 2 - 1
kv = torch.rand(LK * COLS, 2 * INNER, device="cuda", generator=g) * 2 - 1
w_pairs = lib.split_f16((torch.rand(N, INNER, device="cuda", generator=g) * 2 - 1) * INNER ** -0.5, scale=64.0)
out = torch.empty(COLS * LQ, N, device="cuda")
P, Vp = lib.perceiver_probs(q, kv, 1, COLS, LK * COLS, COLS, LQ, LK, HEADS, HD)
for _ in range(3):
    lib.perceiver_out(Vp, w_pairs, P, out, COLS, LQ, LK, HEADS, HD)
torch.cuda.synchronize()
buf = np.zeros(8 * 256, dtype=np.uint32)
lib.load().aurora_hip_debug_po_stamps(ctypes.c_void_p(buf.ctypes.data))
t = buf.reshape(8, 256)[:, :128].astype(np.int64).reshape(8, 32, 4)
t0 = t[0, 0, 0]
for wv in (0, 1, 4, 5):
    print(f"wave {wv}:")
    for st in (0, 1, 2, 8, 9, 10, 11, 30, 31):
        a = t[wv, st]
        nxt = t[wv, st + 1, 0] if st < 31 else a[3]
        print(f"  stage {st:2d}: R start {a[0] - t0:7d} | R {a[1] - a[0]:5d} | barrier {a[2] - a[1]:5d} | C {a[3] - a[2]:5d} | barrier {nxt - a[3]:5d}")
print("stage period (wave 0, stages 8..24):", (t[0, 24, 0] - t[0, 8, 0]) / 16)
