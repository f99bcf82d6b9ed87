"""Build libaurora_hip.so (gfx950) in-tree with hipcc.

    python -m aurora_amd.build [--force]

The library lands in aurora_amd/_lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.
"""

from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "_lib" / "libaurora_hip.so"
SOURCES = ("runtime.hip", "gemm.hip", "gemm_a4.hip", "attention.hip", "norm.hip", "embed.hip", "perceiver_out.hip", "band.hip",
           "model.hip", "step.hip")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale() -> bool:
    if not LIB.exists():
        return True
    built = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.h", CSRC / "band.h", CSRC / "model.h", CSRC / "gemm_a4_loop.inc",
                                          PKG.parent / "include" / "aurora_hip.h"]
    return any(d.stat().st_mtime > built for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> Path:
    """Compile every HIP source for gfx950 and link them into one shared library."""
    if not force and not is_stale():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = LIB.parent / (src.replace(".hip", ".o"))
        # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in arch VGPRs (gfx950 has one unified file);
        # otherwise the softmax in the attention kernel pays a v_accvgpr_read per score.
        # (AURORA_BUILD_FLAGS: extra compiler flags of a probe build, e.g. -DA4_EXPERIMENTS for tools/gemm_a4_stamps.py)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-mllvm",
               "-amdgpu-mfma-vgpr-form=1", *os.environ.get("AURORA_BUILD_FLAGS", "").split(), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout.decode(errors='replace')}")
    for o in objs:
        o.unlink(missing_ok=True)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
