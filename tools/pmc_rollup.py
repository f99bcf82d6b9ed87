"""Roll the per-kernel PMC summary (tools/pmc_summary.py, written by tools/profile_round.sh) up into the figures
bench.py and DESIGN.md quote:

    python tools/pmc_rollup.py gpurun_out/r05_pmc_summary.json profiles/r05_pmc_summary.json $(git rev-parse --short HEAD) gpurun_out/r05_sources.sha256

  linear_bf16_*  launch-weighted means over the two bf16 GEMM kernels (ring and persistent)
  traffic        = 2 x FETCH_SIZE + WRITE_SIZE: FETCH_SIZE is doubled for the GEMM's LDS-DMA pattern as measured by
                   tools/probes/fetch_calib.hip (profiles/r02_fetch_calibration.txt: 0.500 of the bytes for the
                   16-rows-x-64-B pattern, the 8-rows-x-128-B pattern and a contiguous stream alike); WRITE_SIZE is
                   taken at face value -- it reproduces the algorithmic output bytes of the persistent kernel's 134
                   launches per step to 0.1 % (391 MB).
"""
import hashlib
import json
import sys
from pathlib import Path

src, dst = sys.argv[1], sys.argv[2]
build_commit = sys.argv[3] if len(sys.argv) > 3 else None   # commit of the tree the passes profiled (gpurun ships no .git)
# `sha256sum` lines written on the GPU box by tools/profile_round.sh (<tag>_sources.sha256): what the passes really ran
box_sources = {}
if len(sys.argv) > 4:
    for line in open(sys.argv[4]).read().splitlines():
        digest, name = line.split()
        box_sources[Path(name).name] = digest
CSRC = Path(__file__).resolve().parents[1] / "aurora_amd" / "csrc"
d = json.load(open(src))
gemm = {k: v for k, v in d.items() if k.startswith("linear_kernel_256p") or k.startswith("linear_kernel_256<unsigned short")}
n = {k: v["launches"]["fetch"] for k, v in gemm.items()}
tot = sum(n.values())
w = lambda key: sum(v[key] * n[k] for k, v in gemm.items()) / tot  # noqa: E731
out = {
    "_about": "rocprofv3 --pmc passes (FETCH_SIZE+GRBM_GUI_ACTIVE, WRITE_SIZE, SQ counters: three separate runs, "
              "tools/profile_round.sh) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`, aggregated per kernel by "
              "tools/pmc_summary.py and rolled up by tools/pmc_rollup.py; values are means per launch.",
    "_calibration": "FETCH_SIZE reports 0.500 of the bytes actually read for all three access patterns probed "
                    "(profiles/r02_fetch_calibration.txt), so reads = 2 x FETCH_SIZE x 1024 B for every kernel here, the GEMM "
                    "included (round 1 took x1 for the GEMM: wrong).  FETCH_SIZE counts L2 misses on the fabric side: "
                    "Infinity-Cache hits are included, so this is L2-miss traffic, an upper bound of HBM traffic.",
    "build_commit": build_commit,
    # bench.py reports `roofline.traffic` only while the GEMM source is the one these passes profiled, and names every kernel
    # source that has changed since (`profile_stale_sources`): a per-kernel figure of this file is evidence for the build
    # whose sources hash to these values, not for whatever is in the tree later
    "gemm_source_sha256": box_sources.get("gemm.hip") or hashlib.sha256((CSRC / "gemm.hip").read_bytes()).hexdigest(),
    "source_sha256": box_sources or {f.name: hashlib.sha256(f.read_bytes()).hexdigest()
                                     for f in sorted(CSRC.iterdir()) if f.suffix in (".hip", ".h")},
    "source_sha256_recorded": "on the GPU box, before the passes" if box_sources else "at roll-up time, from the tree",
    "linear_bf16_launches_profiled": tot,
    "linear_bf16_read_bytes_per_launch": 2.0 * w("FETCH_SIZE_per_launch") * 1024,
    "linear_bf16_write_bytes_per_launch": w("WRITE_SIZE_per_launch") * 1024,
    # MFMA-busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy / (GRBM / 8 * 1024)
    "linear_bf16_mfma_busy_fraction": w("SQ_VALU_MFMA_BUSY_CYCLES_per_launch") / (w("GRBM_GUI_ACTIVE_per_launch") * 128),
}
out["linear_bf16_hbm_bytes_per_launch"] = out["linear_bf16_read_bytes_per_launch"] + out["linear_bf16_write_bytes_per_launch"]
att = next(v for k, v in d.items() if k.startswith("window_attention_bf16"))
out["window_attention_bf16_hbm_bytes_per_launch"] = 2.0 * att["FETCH_SIZE_per_launch"] * 1024 + att["WRITE_SIZE_per_launch"] * 1024
out["per_kernel"] = d
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
