"""Per-shape throughput of aurora_hip_linear on the backbone GEMM shapes of the 0.25-degree config."""
import sys
from pathlib import Path

import os

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from aurora_amd.engine import lib  # noqa: E402

SHAPES = [  # (name, M, N, K, weight in the step)
    ("s0.qkv", 259200, 1536, 512, 12), ("s0.proj", 259200, 512, 512, 12),
    ("s0.fc1", 259200, 2048, 512, 12), ("s0.fc2", 259200, 512, 2048, 12),
    ("s1.qkv", 64800, 3072, 1024, 20), ("s1.proj", 64800, 1024, 1024, 20),
    ("s1.fc1", 64800, 4096, 1024, 20), ("s1.fc2", 64800, 1024, 4096, 20),
    ("s2.qkv", 16200, 6144, 2048, 16), ("s2.proj", 16200, 2048, 2048, 16),
    ("s2.fc1", 16200, 8192, 2048, 16), ("s2.fc2", 16200, 2048, 8192, 16),
    ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0),
    # per-rank shapes of an 8-way latitude-band split (stage 0: 34560 rows, stage 1: 8100 rows, stage 2: 2160 rows)
    ("r8.s0.qkv", 34560, 1536, 512, 0), ("r8.s0.proj", 34560, 512, 512, 0), ("r8.s0.fc1", 34560, 2048, 512, 0),
    ("r8.s0.fc2", 34560, 512, 2048, 0),
    ("r8.s1.qkv", 8100, 3072, 1024, 0), ("r8.s1.proj", 8100, 1024, 1024, 0), ("r8.s1.fc1", 8100, 4096, 1024, 0),
    ("r8.s1.fc2", 8100, 1024, 4096, 0), ("r8.s2.qkv", 2160, 6144, 2048, 0), ("r8.s2.proj", 2160, 2048, 2048, 0),
    ("r8.s2.fc1", 2160, 8192, 2048, 0), ("r8.s2.fc2", 2160, 2048, 8192, 0),
    # fp32 encoder / decoder linears of the 0.25-degree step (run with `f32`): level aggregation (D = 512, 194,400 latent
    # rows, 842,400 context rows) and level decoder (D = 1024, 842,400 query rows, 194,400 context rows)
    ("d.fc1", 842400, 2048, 1024, 0), ("d.fc2", 842400, 1024, 2048, 0), ("d.out", 842400, 1024, 1024, 0),
    ("d.kv", 194400, 2048, 1024, 0), ("e.kv", 842400, 1024, 512, 0), ("e.fc1", 194400, 2048, 512, 0),
    ("e.fc2", 194400, 512, 2048, 0), ("e.embed", 842400, 512, 160, 0),
]
F32_SHAPES = "d.fc1,d.fc2,d.out,d.kv,e.kv,e.fc1,e.fc2"
# GEMM_BENCH_PRESPLIT = w | aw | awc: operands (and the result) in the fp16-pair layout of the two-term kernel
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
if len(sys.argv) > 2:  # restrict to the named shapes
    names = (F32_SHAPES if sys.argv[2] == "f32shapes" else sys.argv[2]).split(",")
    SHAPES = [s_ for s_ in SHAPES if s_[0] in names]
PRE = os.environ.get("GEMM_BENCH_PRESPLIT", "")
FLAGS = (lib.F32_W_SPLIT if "w" in PRE else 0) | (lib.F32_A_SPLIT if "a" in PRE else 0) | (lib.F32_C_SPLIT if "c" in PRE else 0)
ACT = int(os.environ.get("GEMM_BENCH_ACT", "0"))   # 1 = GELU epilogue
# GEMM_BENCH_SECONDS = s: every shape runs back to back for s seconds before (and as) it is timed -- the socket then sits at its power
# cap and the clock the cap allows (profiles/r06_clocks_under_load.log); the default five repetitions on a cool chip flatter every kernel
SUSTAIN = float(os.environ.get("GEMM_BENCH_SECONDS", "0"))
VENDOR = bool(os.environ.get("GEMM_BENCH_VENDOR"))   # also time torch's own linear (hipBLASLt / rocBLAS) as a yardstick
torch.backends.cuda.matmul.allow_tf32 = False
tot_ms = tot_fl = ven_ms = 0.0
for name, M, N, K, wt in SHAPES:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(dtype)
    if os.environ.get("GEMM_BENCH_ZERO"):   # power experiment: all-zero operands toggle no datapath bits
        a.zero_()
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).to(dtype)
    if os.environ.get("GEMM_BENCH_ZERO"):
        w.zero_()
    b = torch.rand(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    if FLAGS & lib.F32_W_SPLIT:
        w = lib.split_f16(w, scale=64.0)
    if FLAGS & lib.F32_A_SPLIT:
        a = lib.split_f16(a)
    for _ in range(2):
        lib.linear(a, w, b, out, act=ACT, presplit=FLAGS)
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if SUSTAIN > 0:   # calibrate, then one timed run of ~SUSTAIN seconds
        e0.record()
        for _ in range(20):
            lib.linear(a, w, b, out, act=ACT, presplit=FLAGS)
        e1.record()
        torch.cuda.synchronize()
        reps = max(20, int(SUSTAIN * 1e3 / (e0.elapsed_time(e1) / 20)))
    e0.record()
    for _ in range(reps):
        lib.linear(a, w, b, out, act=ACT, presplit=FLAGS)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * N * K
    tot_ms += ms * wt
    tot_fl += fl * wt
    vendor = ""
    if VENDOR and FLAGS:
        vendor = "   | (vendor yardstick skipped: pre-split operands have no vendor counterpart)"
    if VENDOR and not FLAGS:
        # YARDSTICK ONLY (never part of the product path): the same product through the vendor library torch dispatches to
        # (hipBLASLt / rocBLAS), same operands, bias in the call, GELU as a second kernel where asked for; fp32 = true fp32
        # (allow_tf32 off, which is torch's default), i.e. what the reference's fp32 encoder / decoder would run on this GPU.
        # (the bias is converted ONCE, outside the timed loop; GELU stays a separate launch for the vendor leg -- our epilogue
        #  fuses it --, so with ACT = 1 the ratio flatters us by that launch: the yardstick's headline runs are ACT = 0)
        act = torch.nn.functional.gelu if ACT == 1 else (lambda t: t)
        bt = b.to(dtype)
        for _ in range(2):
            act(torch.nn.functional.linear(a, w, bt))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):   # (the same number of repetitions as ours: sustained when GEMM_BENCH_SECONDS is set)
            act(torch.nn.functional.linear(a, w, bt))
        e1.record()
        torch.cuda.synchronize()
        vms = e0.elapsed_time(e1) / reps
        ven_ms += vms * wt
        vendor = f"   | vendor library via torch {vms:8.3f} ms  {fl / vms / 1e9:8.1f} TFLOP/s  (x{vms / ms:.2f} of ours)"
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s{vendor}", flush=True)
    del a, w, out
if tot_ms:
    print(f"weighted backbone: {tot_ms:.1f} ms/step, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
    if ven_ms:
        print(f"vendor library, same weights: {ven_ms:.1f} ms/step, {tot_fl / ven_ms / 1e9:.1f} TFLOP/s")
