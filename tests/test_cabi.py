"""The C-ABI library builds/loads here (no GPU) and exports every symbol include/aurora_hip.h declares."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built():
    from aurora_amd.build import build_library

    return build_library(force=False, verbose=False)


def test_header_symbols_are_exported(built):
    header = (ROOT / "include" / "aurora_hip.h").read_text()
    declared = set(re.findall(r"\b(aurora_hip_\w+)\s*\(", header))
    assert len(declared) >= 12
    lib = ctypes.CDLL(str(built))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    from aurora_amd.engine import lib as shim

    assert set(shim.EXPORTED_SYMBOLS) == declared
    assert shim.load().aurora_hip_version() >= 1


def test_ctypes_structs_match_header_layout():
    from aurora_amd.engine import lib as shim

    # field-for-field with the typedefs in include/aurora_hip.h
    assert [f[0] for f in shim.PatchVar._fields_] == [
        "src", "stride_b", "stride_t", "stride_c", "stride_h", "stride_w", "loc", "inv_scale",
        "transform", "tw0", "tw1", "tb"]
    assert ctypes.sizeof(shim.PatchVar) == 80
    assert [f[0] for f in shim.UnpatchVar._fields_] == [
        "dst", "loc", "scale", "clamp_min0", "col0", "lvl_stride", "mod_col0", "prev", "prev_sb",
        "prev_sc", "prev_sh", "inv_scale", "clamp_max1_levels", "angle_col0", "dens_col0", "mask", "mask_sh",
        "mask_thresh"]
    assert ctypes.sizeof(shim.UnpatchVar) == 120
    # ... and byte for byte with what the compiler made of every struct of the header
    sizes = (ctypes.c_int32 * 16)()
    n = shim.load().aurora_hip_abi_sizes(sizes, 16)
    mine = [shim.HipConfig, shim.HipGrid, shim.HipStepIO, shim.HipBand, shim.HipHaloMsg, shim.HipPlanInfo, shim.PatchVar,
            shim.UnpatchVar, shim.HipProfileEntry]
    assert n == len(mine)
    assert [ctypes.sizeof(t) for t in mine] == list(sizes[:n])


def test_argument_errors_surface_without_a_gpu(built):
    """Argument validation happens before any launch, so it is testable on the CPU box."""
    lib = ctypes.CDLL(str(built))
    lib.aurora_hip_last_error.restype = ctypes.c_char_p
    from aurora_amd.engine import lib as shim

    fn = shim.load().aurora_hip_linear
    code = fn(16, 40, 16, 40, None, 16, 8, None, 0, None, 0, 4, 8, 40, 0, 0, None)  # K=40 fp32: not 32-multiple
    assert code == -1 and b"multiple" in shim.load().aurora_hip_last_error()
    code = shim.load().aurora_hip_window_attention(16, None, 16, 16, None, 1, 10, 10, 96, 2, 1, 4, 1, None)
    assert code == -1 and b"head_dim" in shim.load().aurora_hip_last_error()


def test_constants_and_new_argument_contracts_match_the_header(built):
    """Flag values and profile-table length of the Python shim equal the header's; the argument contracts of the
    round-2 entry points (pre-split operands, fused linear + LayerNorm, pair-layout LayerNorm) fail before any launch."""
    from aurora_amd.engine import lib as shim

    header = (ROOT / "include" / "aurora_hip.h").read_text()
    defines = {k: int(v) for k, v in re.findall(r"#define (AURORA_F32_[ACW]_SPLIT) (\d+)", header)}
    assert defines == {"AURORA_F32_A_SPLIT": shim.F32_A_SPLIT, "AURORA_F32_W_SPLIT": shim.F32_W_SPLIT,
                       "AURORA_F32_C_SPLIT": shim.F32_C_SPLIT}
    assert f"capacity >= {len(shim.PROFILE_KINDS)}" in header
    for kind in shim.PROFILE_KINDS:
        assert kind in header, kind
    L = shim.load()
    err = lambda: L.aurora_hip_last_error()  # noqa: E731
    # pre-split operands: two-term mode only, N % 256 == 0, K % 32 == 0, K >= 96; a pre-split A needs a pre-split W
    args = lambda N, K, mode: (16, K, 16, K, None, 16, N, None, 0, None, 0, 512, N, K, 0, 0, mode, None, 0.0, None)  # noqa: E731
    assert L.aurora_hip_linear_ex(*args(512, 128, 2 | shim.F32_A_SPLIT)) == -1 and b"pre-split weights" in err()
    assert L.aurora_hip_linear_ex(*args(512, 64, 2 | shim.F32_W_SPLIT)) == -1 and b"fp16-pair" in err()
    assert L.aurora_hip_linear_ex(*args(80, 128, 2 | shim.F32_W_SPLIT)) == -1 and b"fp16-pair" in err()
    assert L.aurora_hip_linear_ex(*args(512, 128, 1 | shim.F32_W_SPLIT)) == -1
    # fused linear + LayerNorm: rows of 512 only
    assert L.aurora_hip_linear_layernorm(16, 512, 16, 512, None, None, None, 16, 1024, 16, 1024, None, 0, 256, 1024, 512,
                                         1e-5, None) == -1 and b"512" in err()
    # pair-layout LayerNorm: D and the pair strides in multiples of 32
    assert L.aurora_hip_layernorm_split(16, 48, None, None, None, 0, 0, 0, None, 0, 16, 48, 4, 48, 1e-5, None) == -1
    assert L.aurora_hip_split_f16(16, 48, 16, 48, 4, 48, 1.0, None) == -1 and b"32" in err()
    assert L.aurora_hip_perceiver_attention_ex(16, 0, 16, 16, 1, 4, 4, 1, 3, 3, 3, 16, 1, 16, 1.0, None) == -1
    # round 4 -- split-K with lent scratch: the plan is host arithmetic (the per-rank shapes of an 8-way band at the coarsest
    # stage want three K-slices of their 72 tiles, the un-sharded shapes none), the scratch must be 16-byte aligned
    assert L.aurora_hip_linear_workspace(2160, 2048, 8192, shim.BF16) == 72 * 3 * 256 * 256 * 4
    assert L.aurora_hip_linear_workspace(2160, 2048, 2048, shim.BF16) == 0      # a slice would keep < 64 K-steps
    assert L.aurora_hip_linear_workspace(16200, 2048, 8192, shim.BF16) == 0     # 512 tiles: nothing idles
    assert L.aurora_hip_linear_workspace(2160, 2048, 8192, shim.F32) == 0       # bf16 only
    assert L.aurora_hip_linear_ws(16, 8192, 16, 8192, None, 16, 2048, None, 0, None, 0, 2160, 2048, 8192, shim.BF16, 0,
                                  8, 1 << 26, 16, 4096, 0, None) == -1 and b"workspace" in err()
    assert L.aurora_hip_zero_words(None, 4, None) == -1 and L.aurora_hip_zero_words(16, 65, None) == -1
