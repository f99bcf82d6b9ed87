"""Which kernels the vendor library (hipBLASLt / rocBLAS through torch) picks for the backbone's long-K GEMM shapes: run under
`rocprofv3 --kernel-trace --stats`; the Tensile kernel names spell out macro tile, wave tile, depthU, LDS and stream-K settings.
YARDSTICK ONLY -- nothing in aurora_amd/ calls a vendor GEMM."""
import sys

import torch

SHAPES = [("s0.qkv", 259200, 1536, 512), ("s1.fc1", 64800, 4096, 1024), ("s1.fc2", 64800, 1024, 4096),
          ("s2.qkv", 16200, 6144, 2048), ("s2.fc2", 16200, 2048, 8192), ("sq8192", 8192, 8192, 8192)]
for name, M, N, K in SHAPES:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) * K ** -0.5).bfloat16()
    b = torch.rand(N, device="cuda").bfloat16()
    for _ in range(4):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    print(name, "done", file=sys.stderr)
