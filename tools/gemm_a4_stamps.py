"""Cycles per MFMA of the hand-scheduled loop of linear_kernel_256a4 (csrc/gemm_a4.hip) from its own s_memtime stamps, next to
the launch's wall-clock rate and the eight-wave ping-pong kernel's on the same operands; results compared bit for bit.
    AURORA_GEMM_A4_MIN_K is set by this script per leg (separate processes: the default is read once)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SHAPES = [("s1.qkv", 64800, 3072, 1024), ("s1.fc1", 64800, 4096, 1024), ("s1.fc2", 64800, 1024, 4096), ("s2.qkv", 16200, 6144, 2048),
          ("s2.fc1", 16200, 8192, 2048), ("s2.fc2", 16200, 2048, 8192), ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]

if len(sys.argv) > 1 and sys.argv[1] == "leg":
    if os.environ.get("A4_SHAPES"):
        SHAPES = [s_ for s_ in SHAPES if s_[0] in os.environ["A4_SHAPES"].split(",")]
    import torch
    sys.path.insert(0, str(ROOT))
    from aurora_amd.engine import lib
    a4 = int(os.environ["AURORA_GEMM_A4_MIN_K"]) > 0
    stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
    if a4:
        lib.load().aurora_hip_debug_a4_stamps(stamps.data_ptr())
    for name, M, N, K in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).bfloat16()
        b = torch.rand(N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(2):
            lib.linear(a, w, b, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.linear(a, w, b, out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        cyc = ""
        if a4:
            s = stamps.cpu().reshape(8, 8)
            per = [(int(r[2]) - int(r[1])) / (128.0 * int(r[3])) for r in s if int(r[3]) > 0]
            pro = [int(r[1]) - int(r[0]) for r in s if int(r[3]) > 0]
            epi = [int(r[4]) - int(r[2]) for r in s if int(r[3]) > 0]
            tot = [int(r[4]) - int(r[0]) for r in s if int(r[3]) > 0]
            cyc = (f" loop {min(per):.2f}-{max(per):.2f} cycles/MFMA (8 waves of 2 workgroups), prologue {min(pro)}-{max(pro)}, "
                   f"epilogue {min(epi)}-{max(epi)}, tile {min(tot)}-{max(tot)} cycles")
        chk = int(out.view(torch.int16).to(torch.int64).sum().item())
        print(f"{name:8s} {ms:8.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s checksum {chk}{cyc}", flush=True)
    sys.exit(0)

legs = [("eight-wave ping-pong", "0", "0")] + [(f"four-wave hand-scheduled, variant {v}", "64", v) for v in (sys.argv[1:] or ["0"])]
for label, mink, variant in legs:
    print(f"# {label}", flush=True)
    subprocess.run([sys.executable, __file__, "leg"], env={**os.environ, "AURORA_GEMM_A4_MIN_K": mink, "AURORA_GEMM_A4_VARIANT": variant},
                   check=False)
