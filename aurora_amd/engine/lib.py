"""ctypes binding of libaurora_hip.so (include/aurora_hip.h).

PyTorch is used for device memory and streams only: every wrapper below takes torch CUDA
(HIP) tensors, checks what the C ABI cannot check (device, dtype, contiguity of the inner
dimension) and passes raw device pointers plus the current HIP stream.  A missing library is a
hard error -- there is no fallback implementation in this package.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_float, c_int, c_int32, c_int64, c_void_p
from pathlib import Path
from typing import Optional

import torch

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2

_LIB_PATH = Path(__file__).resolve().parents[1] / "_lib" / "libaurora_hip.so"


class PatchVar(ctypes.Structure):
    _fields_ = [
        ("src", c_void_p), ("stride_b", c_int64), ("stride_t", c_int64), ("stride_c", c_int64),
        ("stride_h", c_int64), ("stride_w", c_int64), ("loc", c_void_p), ("inv_scale", c_void_p),
        ("transform", c_int32), ("tw0", c_float), ("tw1", c_float), ("tb", c_float),
    ]


class UnpatchVar(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("loc", c_void_p), ("scale", c_void_p),
                ("clamp_min0", c_int32), ("col0", c_int32), ("lvl_stride", c_int32), ("mod_col0", c_int32),
                ("prev", c_void_p), ("prev_sb", c_int64), ("prev_sc", c_int64), ("prev_sh", c_int64),
                ("inv_scale", c_void_p), ("clamp_max1_levels", ctypes.c_uint32),
                ("angle_col0", c_int32), ("dens_col0", c_int32), ("mask", c_void_p), ("mask_sh", c_int64),
                ("mask_thresh", c_float)]


def unpatch_var(dst: int, loc: int, scale: int, clamp_min0: int, col0: int) -> UnpatchVar:
    """An `aurora_unpatch_var` with every optional feature switched off."""
    d = UnpatchVar(dst, loc, scale, clamp_min0, col0)
    d.mod_col0 = d.angle_col0 = d.dens_col0 = -1
    return d


_SIGNATURES = {
    "aurora_hip_version": (c_int, []),
    "aurora_hip_last_error": (ctypes.c_char_p, []),
    "aurora_hip_default_f32_gemm": (c_int, []),
    "aurora_hip_linear_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                     c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p, c_float, c_void_p]),
    "aurora_hip_linear_workspace": (c_int64, [c_int64, c_int, c_int, c_int]),
    "aurora_hip_linear_ws": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                     c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p,
                                     c_int, c_int, c_void_p]),
    "aurora_hip_linear_batched": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                          c_int, c_int, c_int, c_int, c_void_p, c_float, c_int, c_int64, c_int64, c_int64,
                                          c_int64, c_void_p]),
    "aurora_hip_absmax": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "aurora_hip_absmax_fold": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "aurora_hip_zero_words": (c_int, [c_void_p, c_int, c_void_p]),
    "aurora_hip_linear_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_float,
                                            c_void_p]),
    "aurora_hip_split_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]),
    "aurora_hip_layernorm_split": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                           c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]),
    "aurora_hip_linear": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    "aurora_hip_window_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                            c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "aurora_hip_window_attention_planes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "aurora_hip_linear_planes": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int64,
                                         c_int, c_int, c_int, c_void_p]),
    "aurora_hip_gather_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                       c_void_p]),
    "aurora_hip_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                     c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_float,
                                     c_int, c_void_p]),
    "aurora_hip_merge_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_float, c_int, c_void_p]),
    "aurora_hip_split_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "aurora_hip_patchify": (c_int, [ctypes.POINTER(PatchVar), c_int, c_void_p, c_int64, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "aurora_hip_patchify_absmax": (c_int, [ctypes.POINTER(PatchVar), c_int, c_void_p, c_int64, c_int, c_int,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "aurora_hip_perceiver_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64,
                                               c_int64, c_int64, c_int, c_int, c_int, c_int, c_int,
                                               c_void_p]),
    "aurora_hip_perceiver_attention_ex": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64,
                                                  c_int64, c_int64, c_int, c_int, c_int, c_int, c_int,
                                                  c_void_p, c_float, c_void_p]),
    "aurora_hip_perceiver_attention_unless": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64,
                                                      c_int64, c_int64, c_int, c_int, c_int, c_int, c_int,
                                                      c_void_p, c_float, c_void_p, c_float, c_void_p]),
    "aurora_hip_perceiver_out_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "aurora_hip_perceiver_probs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64,
                                           c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
    "aurora_hip_perceiver_attention_scores": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int64, c_int64,
                                                      c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_float,
                                                      c_void_p]),
    "aurora_hip_perceiver_probs_scores": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                                  c_int64, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
    "aurora_hip_perceiver_out": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
    "aurora_hip_assemble_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "aurora_hip_unpatchify": (c_int, [c_void_p, c_int64, ctypes.POINTER(UnpatchVar), c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    "aurora_hip_copy2d": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int,
                                  c_void_p]),
    "aurora_hip_convert": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[ctypes.CDLL] = None


def library_path() -> Path:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the HIP library once; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m aurora_amd.build` "
                "(hipcc, gfx950). aurora_amd has no fallback compute path."
            )
        # (kernel experiments: AURORA_HIP_LIB points at an alternative build of the same ABI)
        lib = ctypes.CDLL(os.environ.get("AURORA_HIP_LIB") or str(_LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class HipError(RuntimeError):
    pass


def _check(code: int) -> None:
    if code != 0:
        msg = load().aurora_hip_last_error().decode(errors="replace")
        # -1 = AURORA_E_ARG: the Python shim re-raises argument violations as the assertion /
        # value errors the reference raises for bad shapes.
        raise (ValueError if code == -1 else HipError)(f"libaurora_hip: {msg} (code {code})")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported compute dtype {dt}: the HIP engine computes in fp32 or bf16")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "tensor must live on the HIP device"
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rows(t: torch.Tensor) -> tuple[int, int]:
    """(leading dimension in elements, inner size) of a 2-D view with unit inner stride."""
    assert t.dim() == 2 and t.stride(1) == 1, f"need a row-major 2-D view, got {t.shape} / {t.stride()}"
    if t.shape[0] == 1:  # the stride of a size-1 dimension is arbitrary (often 0): irrelevant here
        return max(t.stride(0), t.shape[1]), t.shape[1]
    return t.stride(0), t.shape[1]


# ---- optional per-launch timing (bench.py / tools): HIP events on the launch stream -----------
_profile: Optional[list] = None
_profile_only: Optional[set] = None


def profile_start(only: Optional[set] = None) -> None:
    """Start recording (kernel, algorithmic work, start event, stop event) for every launch, or for the
    kernels named in `only` (an event pair around a launch keeps it from overlapping its neighbours, so timing
    every launch of a step costs a few percent of the step)."""
    global _profile, _profile_only
    _profile = []
    _profile_only = only


def profile_stop() -> dict:
    """Stop recording; returns {kernel: {"launches", "ms", "work"}} (synchronises the device)."""
    global _profile
    rec, _profile = _profile or [], None
    torch.cuda.synchronize()
    out: dict = {}
    for name, work, e0, e1 in rec:
        d = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0})
        d["launches"] += 1
        d["ms"] += e0.elapsed_time(e1)
        d["work"] += work
    return out


class _Timed:
    """Brackets one launch with HIP events when profiling is on (torch.cuda.Event records on the
    current stream, which is the stream every kernel of this library is launched on)."""

    def __init__(self, name: str, work: float):
        self.name, self.work = name, work

    def __enter__(self):
        self.on = _profile is not None and (_profile_only is None or self.name in _profile_only)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            _profile.append((self.name, self.work, self.e0, self.e1))
        return False


# ---- wrappers ------------------------------------------------------------------------------
# How fp32 linears issued by THIS thread are multiplied: (mode, guard tensor, guard limit); mode -1 = process default.
# Passed per call to aurora_hip_linear_ex -- the library itself keeps no mutable state, so two engines on different
# threads / streams cannot disturb each other, and nothing hidden is baked into a captured graph.
_f32 = threading.local()


def _f32_state():
    return getattr(_f32, "state", (-1, None, 0.0))


def default_f32_gemm() -> int:
    """The process default (AURORA_F32_GEMM=native|bf16|f16): 0 native fp32 MFMA, 1 three bf16 terms, 2 two fp16 terms."""
    return load().aurora_hip_default_f32_gemm()


class f32_gemm:
    """`with f32_gemm(mode):` -- fp32 linears issued by this thread inside use `mode` (0 native fp32 MFMA, 1 exact
    3 x bf16 operand splitting, 2 2 x fp16 splitting)."""

    def __init__(self, mode: int, guard=None):
        self.state = (mode, None if guard is None else guard[0], 0.0 if guard is None else float(guard[1]))

    def __enter__(self):
        self.prev = _f32_state()
        _f32.state = self.state

    def __exit__(self, *exc):
        _f32.state = self.prev
        return False


def absmax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max |x| of a contiguous fp32 tensor into a one-element device tensor (no host synchronisation)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device) if out is None else out
    with _Timed("absmax", 0.0):
        _check(load().aurora_hip_absmax(_ptr(x), x.numel(), _ptr(out), _stream()))
    return out


class bounded_activations(f32_gemm):
    """`with bounded_activations():` -- the fp32 linears issued inside may use the 2 x fp16 operand split (three
    MFMAs instead of six, the same 2^-24 operand accuracy).  Without arguments the caller vouches that their
    activation operand is bounded by construction -- a LayerNorm output or the GELU of a linear of one -- i.e. far
    inside fp16's range (|x| < 65504) whatever the model's inputs are.  With `guard=(amax, limit)` the decision is
    taken on the device, per launch: fp16 terms iff `amax[0] < limit` (`amax` from `absmax`, or a bound derived from
    it), three bf16 terms otherwise.  Honours an explicit native / bf16 choice made through AURORA_F32_GEMM or an
    enclosing `f32_gemm(0)`."""

    def __init__(self, guard=None):
        super().__init__(2, guard)

    def __enter__(self):
        self.prev = _f32_state()
        explicit = self.prev[0] >= 0 or os.environ.get("AURORA_F32_GEMM") is not None
        _f32.state = self.prev if explicit else self.state


F32_A_SPLIT, F32_W_SPLIT, F32_C_SPLIT = 4, 8, 16   # include/aurora_hip.h: operands / output in the fp16-pair layout


def two_term_free() -> bool:
    """True unless the user pinned an fp32 GEMM mode (AURORA_F32_GEMM or an enclosing `f32_gemm`) -- the condition under
    which `bounded_activations` switches to the two-term split, and so the one for handing it pre-split operands."""
    return _f32_state()[0] < 0 and os.environ.get("AURORA_F32_GEMM") is None


def presplit_ok(n: int, k: int) -> bool:
    """Shapes the pre-split form of the two-term kernel takes (include/aurora_hip.h)."""
    return n % 256 == 0 and k % 32 == 0 and k >= 96


def split_f16(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 rows -> the fp16-pair layout of the two-term GEMMs (same shape, dtype float32 as a container: per 32
    features 32 high halves, then 32 remainders).  Weights take scale = 64."""
    assert x.dtype == torch.float32 and x.dim() == 2
    ld, K = _rows(x)
    out = torch.empty_like(x, memory_format=torch.contiguous_format) if out is None else out
    ldo, _ = _rows(out)
    _check(load().aurora_hip_split_f16(_ptr(x), ld, _ptr(out), ldo, x.shape[0], K, scale, _stream()))
    return out


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *,
           out2: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: int = ACT_NONE, n: Optional[int] = None, k: Optional[int] = None, presplit: int = 0) -> torch.Tensor:
    """out[M, N] = act(a[M, K] @ w[N, K].T + bias) (+ residual); all 2-D row-major views.
    `presplit`: F32_* flags -- which of a / w / out are in the fp16-pair layout (two-term mode only)."""
    lda, ka = _rows(a)
    ldw, kw = _rows(w)
    K = k if k is not None else ka
    N = n if n is not None else w.shape[0]
    M = a.shape[0]
    assert kw >= K and ka >= K and a.dtype == w.dtype == out.dtype, (a.dtype, w.dtype, out.dtype)
    assert out.shape[0] == M and out.shape[1] >= N
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() >= N and bias.is_contiguous())
    ldc, _ = _rows(out)
    ldc2 = ldr = 0
    if out2 is not None:
        assert out2.dtype != out.dtype and out2.shape[0] == M
        ldc2, _ = _rows(out2)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape[0] == M
        ldr, _ = _rows(residual)
    name = "linear_bf16" if a.dtype == torch.bfloat16 else "linear_f32"
    with _Timed(name, 2.0 * M * N * K):  # algorithmic FLOPs
        mode, guard, limit = _f32_state()
        if presplit:
            mode = 2 | presplit
        _check(load().aurora_hip_linear_ex(_ptr(a), lda, _ptr(w), ldw, _ptr(bias), _ptr(out), ldc, _ptr(out2),
                                           ldc2, _ptr(residual), ldr, M, N, K, dtype_code(a.dtype), act,
                                           mode, _ptr(guard), limit, _stream()))
    return out


def linear_workspace(M: int, N: int, K: int, dtype: torch.dtype = torch.bfloat16) -> int:
    """Bytes of scratch `linear_ws` would like for this shape (0: the library would not split it along K)."""
    return int(load().aurora_hip_linear_workspace(M, N, K, dtype_code(dtype)))


def linear_ws(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, workspace: torch.Tensor,
              tickets: torch.Tensor, *, split: int = 0, out2: Optional[torch.Tensor] = None,
              residual: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """`linear` with lent scratch: few-tile / long-K bf16 problems are split along K inside one launch (aurora_hip_linear_ws).
    `workspace`: any contiguous tensor; `tickets`: int32, zero on entry, left zero.  split = 0 lets the library choose."""
    lda, K = _rows(a)
    ldw, _ = _rows(w)
    M, N = a.shape[0], w.shape[0]
    assert a.dtype == w.dtype == out.dtype and tickets.dtype == torch.int32 and workspace.is_contiguous()
    ldc, _ = _rows(out)
    ldc2 = ldr = 0
    if out2 is not None:
        ldc2, _ = _rows(out2)
    if residual is not None:
        ldr, _ = _rows(residual)
    with _Timed("linear_bf16" if a.dtype == torch.bfloat16 else "linear_f32", 2.0 * M * N * K):
        _check(load().aurora_hip_linear_ws(_ptr(a), lda, _ptr(w), ldw, _ptr(bias), _ptr(out), ldc, _ptr(out2), ldc2,
                                           _ptr(residual), ldr, M, N, K, dtype_code(a.dtype), act, _ptr(workspace),
                                           workspace.numel() * workspace.element_size(), _ptr(tickets), tickets.numel(), split,
                                           _stream()))
    return out


def linear_layernorm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], gain: Optional[torch.Tensor],
                     shift: Optional[torch.Tensor], x: torch.Tensor, x_out: torch.Tensor,
                     x_bf16: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """x_out = x + LN(a @ w.T + bias) * gain + shift, x_bf16 = bf16(x_out), in one launch (bf16 a / w, N = 512)."""
    lda, K = _rows(a)
    ldw, _ = _rows(w)
    M, N = a.shape[0], w.shape[0]
    assert a.dtype == w.dtype == torch.bfloat16 and x.dtype == x_out.dtype == torch.float32
    ldx, _ = _rows(x)
    ldo, _ = _rows(x_out)
    ldb = 0
    if x_bf16 is not None:
        assert x_bf16.dtype == torch.bfloat16
        ldb, _ = _rows(x_bf16)
    with _Timed("linear_layernorm_bf16", 2.0 * M * N * K):
        _check(load().aurora_hip_linear_layernorm(_ptr(a), lda, _ptr(w), ldw, _ptr(bias), _ptr(gain), _ptr(shift), _ptr(x),
                                                  ldx, _ptr(x_out), ldo, _ptr(x_bf16), ldb, M, N, K, eps, _stream()))
    return x_out


def linear_planes(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, sel0: int = 0) -> torch.Tensor:
    """out[h, m, sel0 + sel, :] = (a @ w.T + bias)[m, (sel * heads + h) * 64 : +64]: `out` is (heads, rows >= M, 3, 64) bf16 --
    q | k | v of a token next to each other, one attention head per plane.  `sel0` = 1: w holds the k | v rows only."""
    M, K = a.shape
    N = w.shape[0]
    heads = out.shape[0]
    assert a.dtype == w.dtype == out.dtype == torch.bfloat16 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == K
    assert out.dim() == 4 and out.shape[2:] == (3, 64) and out.shape[1] >= M and N == (3 - sel0) * 64 * heads
    assert out.stride(3) == 1 and out.stride(2) == 64 and out.stride(1) == 192   # (a row range of longer planes is fine)
    c = out[0, 0, sel0]
    with _Timed("linear_bf16", 2.0 * M * N * K):
        _check(load().aurora_hip_linear_planes(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(c), out.stride(0), heads,
                                               M, N, K, dtype_code(a.dtype), _stream()))
    return out


def window_attention(qkv: torch.Tensor, qkv_bias: Optional[torch.Tensor], out: torch.Tensor,
                     tok: torch.Tensor, grp: Optional[torch.Tensor], B: int, L: int, D: int,
                     heads: int, L_out: Optional[int] = None, planes: bool = False) -> torch.Tensor:
    """`L` rows of qkv per batch element ([owned | halo] for a latitude band), `L_out` rows of out.  `planes`: qkv is
    (heads, B * L, 3, 64), what `linear_planes` writes."""
    L_out = L if L_out is None else L_out
    if planes:
        assert qkv.is_contiguous() and qkv.shape == (heads, B * L, 3, 64) and qkv.dtype == torch.bfloat16
        assert out.numel() == B * L_out * D and out.dtype == qkv.dtype and tok.dtype == torch.int32 and tok.dim() == 2
        n_windows, n_tok = tok.shape
        with _Timed("window_attention_bf16", 4.0 * B * n_windows * n_tok * D * 2):
            _check(load().aurora_hip_window_attention_planes(_ptr(qkv), qkv.stride(0), _ptr(qkv_bias), _ptr(out), _ptr(tok), _ptr(grp),
                                                             B, L, L_out, D, heads, n_windows, n_tok, dtype_code(qkv.dtype),
                                                             _stream()))
        return out
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.numel() == B * L * 3 * D
    assert out.numel() == B * L_out * D and out.dtype == qkv.dtype
    assert tok.dtype == torch.int32 and tok.is_contiguous() and tok.dim() == 2
    assert grp is None or (grp.dtype == torch.uint8 and grp.shape == tok.shape and grp.is_contiguous())
    assert qkv_bias is None or (qkv_bias.dtype == torch.float32 and qkv_bias.numel() == 3 * D)
    n_windows, n_tok = tok.shape
    name = "window_attention_bf16" if qkv.dtype == torch.bfloat16 else "window_attention_f32"
    # algorithmic bytes: q, k, v read + o written once over the padded windows
    with _Timed(name, 4.0 * B * n_windows * n_tok * D * qkv.element_size()):
        _check(load().aurora_hip_window_attention(_ptr(qkv), _ptr(qkv_bias), _ptr(out), _ptr(tok), _ptr(grp),
                                                  B, L, L_out, D, heads, n_windows, n_tok,
                                                  dtype_code(qkv.dtype), _stream()))
    return out


def layernorm(y: torch.Tensor, gain: Optional[torch.Tensor], shift: Optional[torch.Tensor], *,
              res: Optional[torch.Tensor] = None, res_mod: int = 0,
              out_f32: Optional[torch.Tensor] = None, out_t: Optional[torch.Tensor] = None,
              eps: float = 1e-5, d: Optional[int] = None, split_t: bool = False, split_res: bool = False) -> None:
    """`split_t` / `split_res` (fp32 rows only, aurora_hip_layernorm_split): `out_t` receives the fp16-pair layout /
    `res` is in it."""
    ldy, dy = _rows(y)
    D = d if d is not None else dy
    M = y.shape[0]
    for v in (gain, shift):
        assert v is None or (v.dtype == torch.float32 and v.numel() >= D and v.is_contiguous())
    ldr = ldo = ldt = 0
    if res is not None:
        assert res.dtype == torch.float32
        ldr, _ = _rows(res)
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and out_f32.shape[0] == M
        ldo, _ = _rows(out_f32)
    if out_t is not None:
        assert out_t.dtype == y.dtype and out_t.shape[0] == M
        ldt, _ = _rows(out_t)
    nbytes = M * D * (y.element_size() + (4 if res is not None else 0) + (4 if out_f32 is not None else 0)
                      + (y.element_size() if out_t is not None else 0))
    if split_t or split_res:
        assert y.dtype == torch.float32 and (out_t is not None) == split_t and (res is not None or not split_res)
        with _Timed("layernorm", float(nbytes)):
            _check(load().aurora_hip_layernorm_split(_ptr(y), ldy, _ptr(gain), _ptr(shift), _ptr(res), ldr, res_mod,
                                                     1 if split_res else 0, _ptr(out_f32), ldo, _ptr(out_t), ldt, M, D,
                                                     eps, _stream()))
        return
    with _Timed("layernorm", float(nbytes)):
        _check(load().aurora_hip_layernorm(_ptr(y), ldy, _ptr(gain), _ptr(shift), _ptr(res), ldr, res_mod,
                                           _ptr(out_f32), ldo, _ptr(out_t), ldt, M, D, eps,
                                           dtype_code(y.dtype), _stream()))


def merge_ln(x: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, out: torch.Tensor,
             B: int, C: int, H: int, W: int, D: int, eps: float = 1e-5) -> torch.Tensor:
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == B * C * H * W * D
    assert out.is_contiguous() and out.numel() == B * C * ((H + 1) // 2) * ((W + 1) // 2) * 4 * D
    with _Timed("merge_ln", 0.0):
        _check(load().aurora_hip_merge_ln(_ptr(x), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, C, H, W, D, eps,
                                          dtype_code(out.dtype), _stream()))
    return out


def split_ln(y: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, out: torch.Tensor,
             B: int, C: int, H: int, W: int, Dq: int, crop_h: int, crop_w: int,
             eps: float = 1e-5) -> torch.Tensor:
    assert y.is_contiguous() and y.numel() == B * C * H * W * 4 * Dq and y.dtype == out.dtype
    assert out.is_contiguous() and out.numel() == B * C * (2 * H - crop_h) * (2 * W - crop_w) * Dq
    with _Timed("split_ln", 0.0):
        _check(load().aurora_hip_split_ln(_ptr(y), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, C, H, W, Dq,
                                          crop_h, crop_w, eps, dtype_code(y.dtype), _stream()))
    return out


def patchify(desc: list[PatchVar], out: torch.Tensor, k_offset: int, k_total: int, B: int, T: int,
             n_lvl: int, Hp: int, Wp: int, P: int, absmax: Optional[torch.Tensor] = None) -> None:
    """`absmax` (one fp32 word, NOT zeroed here): max |value written| is folded into it."""
    Kpad = out.shape[1]
    assert out.is_contiguous() and out.shape[0] == n_lvl * B * Hp * Wp
    assert absmax is None or (absmax.dtype == torch.float32 and absmax.numel() >= 1)
    arr = (PatchVar * len(desc))(*desc)
    with _Timed("patchify", 0.0):
        _check(load().aurora_hip_patchify_absmax(arr, len(desc), _ptr(out), Kpad, k_offset, k_total, B, T, n_lvl,
                                                 Hp, Wp, P, dtype_code(out.dtype), _ptr(absmax), _stream()))


def perceiver_attention(q: torch.Tensor, q_col_stride: int, kv: torch.Tensor, out: torch.Tensor,
                        B: int, cols_per_b: int, kv_bstride: int, kv_lstride: int, Lq: int, Lk: int,
                        heads: int, head_dim: int, pair_guard=None, skip_guard=None) -> torch.Tensor:
    """`pair_guard=(word, limit)`: fp32 results are written as fp16 pairs iff word[0] < limit (decided on the device).
    `skip_guard=(word, limit)`: the launch retires at once iff word[0] < limit."""
    assert q.is_contiguous() and kv.is_contiguous() and out.is_contiguous()
    assert q.dtype == kv.dtype == out.dtype
    word, limit = pair_guard if pair_guard is not None else (None, 0.0)
    sword, slimit = skip_guard if skip_guard is not None else (None, 0.0)
    with _Timed("perceiver_attention", 0.0):
        _check(load().aurora_hip_perceiver_attention_unless(_ptr(q), q_col_stride, _ptr(kv), _ptr(out), B,
                                                            cols_per_b, kv_bstride, kv_lstride, Lq, Lk, heads,
                                                            head_dim, dtype_code(q.dtype), _ptr(word), float(limit),
                                                            _ptr(sword), float(slimit), _stream()))
    return out


def perceiver_probs(q: torch.Tensor, kv: torch.Tensor, B: int, cols_per_b: int, kv_bstride: int, kv_lstride: int,
                    Lq: int, Lk: int, heads: int, head_dim: int, guard=None):
    """Softmax weights P (n_cols, heads, 64) and the value rows as fp16 pairs Vp (n_cols * Lk, inner) of the
    re-associated decoder attention (aurora_hip_perceiver_probs)."""
    assert q.is_contiguous() and kv.is_contiguous() and q.dtype == kv.dtype == torch.float32
    n_cols, inner = B * cols_per_b, heads * head_dim
    P = torch.zeros((n_cols, heads, 64), device=kv.device, dtype=torch.float32)
    Vp = torch.zeros((n_cols * Lk, inner), device=kv.device, dtype=torch.float32)
    word, limit = guard if guard is not None else (None, 0.0)
    with _Timed("perceiver_attention", 0.0):
        _check(load().aurora_hip_perceiver_probs(_ptr(q), _ptr(kv), _ptr(P), _ptr(Vp), B, cols_per_b, kv_bstride,
                                                 kv_lstride, Lq, Lk, heads, head_dim, _ptr(word), float(limit), _stream()))
    return P, Vp


def perceiver_attention_scores(vs: torch.Tensor, s_off: int, out: torch.Tensor, B: int, cols_per_b: int, kv_bstride: int,
                               kv_lstride: int, Lq: int, Lk: int, heads: int, head_dim: int, pair_guard=None,
                               skip_guard=None) -> torch.Tensor:
    """Perceiver attention from pre-multiplied scores: a row of `vs` is [v | ... | scores (Lq * heads) at s_off]
    (aurora_hip_perceiver_attention_scores)."""
    assert vs.is_contiguous() and out.is_contiguous() and vs.dtype == out.dtype == torch.float32
    word, limit = pair_guard if pair_guard is not None else (None, 0.0)
    sword, slimit = skip_guard if skip_guard is not None else (None, 0.0)
    with _Timed("perceiver_attention", 0.0):
        _check(load().aurora_hip_perceiver_attention_scores(_ptr(vs), vs.shape[1], s_off, _ptr(out), B, cols_per_b, kv_bstride,
                                                            kv_lstride, Lq, Lk, heads, head_dim, _ptr(word), float(limit),
                                                            _ptr(sword), float(slimit), _stream()))
    return out


def perceiver_probs_scores(vs: torch.Tensor, s_off: int, B: int, cols_per_b: int, kv_bstride: int, kv_lstride: int,
                           Lq: int, Lk: int, heads: int, head_dim: int, guard=None):
    """perceiver_probs from pre-multiplied scores (aurora_hip_perceiver_probs_scores)."""
    assert vs.is_contiguous() and vs.dtype == torch.float32
    n_cols, inner = B * cols_per_b, heads * head_dim
    P = torch.zeros((n_cols, heads, 64), device=vs.device, dtype=torch.float32)
    Vp = torch.zeros((n_cols * Lk, inner), device=vs.device, dtype=torch.float32)
    word, limit = guard if guard is not None else (None, 0.0)
    with _Timed("perceiver_attention", 0.0):
        _check(load().aurora_hip_perceiver_probs_scores(_ptr(vs), vs.shape[1], s_off, _ptr(P), _ptr(Vp), B, cols_per_b,
                                                        kv_bstride, kv_lstride, Lq, Lk, heads, head_dim, _ptr(word),
                                                        float(limit), _stream()))
    return P, Vp


def perceiver_out(Vp: torch.Tensor, w_pairs: torch.Tensor, P: torch.Tensor, out: torch.Tensor, n_cols: int, Lq: int,
                  Lk: int, heads: int, head_dim: int, bias: Optional[torch.Tensor] = None, guard=None) -> torch.Tensor:
    """out[col * Lq + l] = sum_h sum_j P[col, h, l, j] W[:, h] Vp[col * Lk + j, h] (aurora_hip_perceiver_out)."""
    assert Vp.is_contiguous() and w_pairs.is_contiguous() and P.is_contiguous() and out.is_contiguous()
    N = out.shape[1]
    word, limit = guard if guard is not None else (None, 0.0)
    with _Timed("perceiver_out", 2.0 * n_cols * Lk * N * heads * head_dim):
        _check(load().aurora_hip_perceiver_out(_ptr(Vp), _ptr(w_pairs), w_pairs.shape[1], _ptr(P), _ptr(bias), _ptr(out),
                                               out.stride(0), n_cols, Lq, Lk, heads, head_dim, N, _ptr(word), float(limit),
                                               _stream()))
    return out


def assemble_tokens(surf: torch.Tensor, agg: torch.Tensor, pos_scale: torch.Tensor,
                    time_emb: torch.Tensor, out_f32: torch.Tensor, out_t: Optional[torch.Tensor],
                    B: int, Cl: int, L: int, D: int) -> None:
    for t in (surf, agg, pos_scale, time_emb, out_f32):
        assert t.dtype == torch.float32 and t.is_contiguous()
    code = BF16 if out_t is not None else F32
    with _Timed("assemble_tokens", 0.0):
        _check(load().aurora_hip_assemble_tokens(_ptr(surf), _ptr(agg), _ptr(pos_scale), _ptr(time_emb),
                                                 _ptr(out_f32), _ptr(out_t), B, Cl, L, D, code, _stream()))


def unpatchify(y: torch.Tensor, desc: list[UnpatchVar], B: int, n_lvl: int, Hp: int, Wp: int,
               P: int) -> None:
    ldy, _ = _rows(y)
    assert y.dtype == torch.float32
    arr = (UnpatchVar * len(desc))(*desc)
    with _Timed("unpatchify", 0.0):
        _check(load().aurora_hip_unpatchify(_ptr(y), ldy, arr, len(desc), B, n_lvl, Hp, Wp, P, _stream()))


def copy2d(src: torch.Tensor, dst: torch.Tensor, cols: Optional[int] = None) -> None:
    lds_, cs = _rows(src)
    ldd, _ = _rows(dst)
    assert src.dtype == dst.dtype and src.shape[0] == dst.shape[0]
    with _Timed("copy2d", 0.0):
        _check(load().aurora_hip_copy2d(_ptr(src), lds_, _ptr(dst), ldd, src.shape[0],
                                        cols if cols is not None else cs, dtype_code(src.dtype), _stream()))


def gather_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[r] = src[idx[r]] for 2-D row-major views with equal row width (bytes multiple of 16)."""
    lds_, w = _rows(src)
    ldd, wd = _rows(dst)
    assert w == wd and src.dtype == dst.dtype and idx.dtype == torch.int32 and idx.is_contiguous()
    assert dst.shape[0] == idx.numel()
    es = src.element_size()
    with _Timed("gather_rows", 0.0):
        _check(load().aurora_hip_gather_rows(_ptr(src), lds_ * es, _ptr(idx), _ptr(dst), ldd * es, idx.numel(), w * es,
                                             _stream()))
    return dst


def convert(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    assert {src.dtype, dst.dtype} == {torch.float32, torch.bfloat16}
    with _Timed("convert", 0.0):
        _check(load().aurora_hip_convert(_ptr(src), _ptr(dst), src.numel(), dtype_code(src.dtype), _stream()))
    return dst


# ---- model handle (one forecast step behind the C ABI) ------------------------------------------------------
_PD = ctypes.POINTER(ctypes.c_double)
_PF = ctypes.POINTER(ctypes.c_float)
_PS = ctypes.POINTER(ctypes.c_char_p)


class HipTuning(ctypes.Structure):   # aurora_hip_config.tuning: 0 = the library's default
    _fields_ = [("fuse_ln", c_int32), ("band_split_attention", c_int32), ("qkv_planes", c_int32), ("split_k", c_int32),
                ("perceiver_reassoc", c_int32), ("score_weights", c_int32), ("reserved", c_int32 * 2)]


def tuning_from_env() -> HipTuning:
    """The AURORA_* switches of INTEGRATION.md, read HERE (once per handle creation) and handed to the library as fields of
    the configuration: the library itself never reads the environment."""
    t = HipTuning()
    e = os.environ.get
    if e("AURORA_FUSE_LN") is not None:
        t.fuse_ln = int(e("AURORA_FUSE_LN")) + 1            # 0 / 1 / 2 -> never / fill rule / always
    for field, var in (("band_split_attention", "AURORA_BAND_SPLIT_ATTENTION"), ("qkv_planes", "AURORA_QKV_PLANES"),
                       ("split_k", "AURORA_SPLIT_K"), ("perceiver_reassoc", "AURORA_PERCEIVER_REASSOC"),
                       ("score_weights", "AURORA_SCORE_WEIGHTS")):
        if e(var) is not None:
            setattr(t, field, 2 if int(e(var)) != 0 else 1)
    return t


class HipConfig(ctypes.Structure):   # aurora_hip_config, field for field
    _fields_ = [("embed_dim", c_int32), ("patch_size", c_int32), ("latent_levels", c_int32), ("num_heads", c_int32),
                ("n_stages", c_int32), ("encoder_depths", c_int32 * 4), ("encoder_heads", c_int32 * 4),
                ("decoder_depths", c_int32 * 4), ("decoder_heads", c_int32 * 4), ("window", c_int32 * 3),
                ("enc_depth", c_int32), ("dec_depth", c_int32), ("perceiver_ln_eps", c_float),
                ("max_history", c_int32), ("timestep_hours", ctypes.c_double), ("stabilise_level_agg", c_int32),
                ("use_lora", c_int32), ("lora_steps", c_int32), ("lora_mode", c_int32), ("autocast", c_int32),
                ("n_surf", c_int32), ("n_static", c_int32), ("n_atmos", c_int32),
                ("surf_vars", _PS), ("static_vars", _PS), ("atmos_vars", _PS),
                # variant keywords
                ("variant", c_int32), ("n_level_condition", c_int32), ("level_condition", _PD),
                ("dynamic_vars", c_int32), ("atmos_static_vars", c_int32), ("clamp_at_first_step", c_int32),
                ("simulate_indexing_bug", c_int32),
                ("n_separate_perceiver", c_int32), ("separate_perceiver", _PS),
                ("n_modulation_heads", c_int32), ("modulation_heads", _PS),
                ("difference_history", ctypes.POINTER(c_int32)),
                ("n_positive_surf", c_int32), ("positive_surf_vars", _PS),
                ("n_positive_atmos", c_int32), ("positive_atmos_vars", _PS),
                ("n_surf_inputs", c_int32), ("surf_inputs", _PS),
                ("n_density", c_int32), ("density_channel_surf_vars", _PS),
                ("n_angle", c_int32), ("angle_surf_vars", _PS), ("tuning", HipTuning)]


class HipHaloMsg(ctypes.Structure):   # aurora_hip_halo_msg
    _fields_ = [("peer", c_int32), ("reserved", c_int32), ("offset", c_int64), ("bytes", c_int64)]


HALO_POST_FN = ctypes.CFUNCTYPE(c_int, c_void_p, ctypes.POINTER(HipHaloMsg), c_int32, ctypes.POINTER(HipHaloMsg), c_int32,
                                c_void_p)
HALO_WAIT_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p)


class HipBand(ctypes.Structure):      # aurora_hip_band
    _fields_ = [("rank", c_int32), ("world", c_int32), ("post", HALO_POST_FN), ("wait", HALO_WAIT_FN), ("user", c_void_p)]


class HipGrid(ctypes.Structure):
    _fields_ = [("n_lat", c_int32), ("n_lon", c_int32), ("lat", _PD), ("lon", _PD), ("n_levels", c_int32),
                ("levels", _PD), ("levels_float32", c_int32), ("surf_loc", _PD), ("surf_scale", _PD),
                ("static_loc", _PD), ("static_scale", _PD), ("atmos_loc", _PD), ("atmos_scale", _PD),
                ("pos_encoding", _PF), ("scale_encoding", _PF)]


class HipStepIO(ctypes.Structure):
    _fields_ = [("B", c_int32), ("T", c_int32), ("surf", ctypes.POINTER(c_void_p)), ("surf_strides", c_int64 * 4),
                ("stat", ctypes.POINTER(c_void_p)), ("static_strides", c_int64 * 2),
                ("atmos", ctypes.POINTER(c_void_p)), ("atmos_strides", c_int64 * 5),
                ("out_surf", ctypes.POINTER(c_void_p)), ("out_atmos", ctypes.POINTER(c_void_p)),
                ("rollout_step", c_int32)]


class HipProfileEntry(ctypes.Structure):
    _fields_ = [("kernel", ctypes.c_char_p), ("launches", c_int64), ("ms", ctypes.c_double), ("work", ctypes.c_double)]


class HipPlanInfo(ctypes.Structure):
    _fields_ = [("n_windows", c_int32), ("win_tokens", c_int32), ("n_own", c_int32), ("n_halo", c_int32),
                ("n_interior", c_int32), ("recv_offset", c_int32 * 2), ("recv_count", c_int32 * 2),
                ("send_count", c_int32 * 2), ("has_groups", c_int32)]


PROFILE_KINDS = ("linear_bf16", "linear_f32", "window_attention_bf16", "layernorm", "merge_ln", "split_ln", "patchify",
                 "perceiver_attention", "assemble_tokens", "unpatchify", "copy2d", "absmax", "linear_layernorm_bf16",
                 "gather_rows", "perceiver_out")

_SIGNATURES.update({
    "aurora_hip_debug_a4_stamps": (None, [c_void_p]),
    "aurora_hip_profile_begin": (c_int, [c_void_p, ctypes.c_uint32]),
    "aurora_hip_profile_end": (c_int, [c_void_p, ctypes.POINTER(HipProfileEntry), c_int, ctypes.POINTER(c_int)]),
    "aurora_hip_profile_end_list": (c_int, [c_void_p, ctypes.POINTER(HipProfileEntry), c_int, ctypes.POINTER(c_int)]),
    "aurora_hip_create": (c_int, [ctypes.POINTER(HipConfig), ctypes.POINTER(c_void_p)]),
    "aurora_hip_destroy": (None, [c_void_p]),
    "aurora_hip_pack_weights": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.POINTER(c_int64), c_int, c_int, c_int]),
    "aurora_hip_finalize": (c_int, [c_void_p, c_void_p]),
    "aurora_hip_save_packed": (c_int, [c_void_p, ctypes.c_char_p, c_void_p]),
    "aurora_hip_load_packed": (c_int, [c_void_p, ctypes.c_char_p]),
    "aurora_hip_precompute": (c_int, [c_void_p, ctypes.POINTER(HipGrid), c_void_p]),
    "aurora_hip_set_time": (c_int, [c_void_p, _PD, c_int, c_void_p]),
    "aurora_hip_step": (c_int, [c_void_p, ctypes.POINTER(HipStepIO), c_void_p]),
    "aurora_hip_workspace_bytes": (c_int64, [c_void_p]),
    "aurora_hip_guard_words": (c_int, [c_void_p, ctypes.POINTER(c_float), c_void_p]),
    "aurora_hip_abi_sizes": (c_int, [ctypes.POINTER(c_int32), c_int]),
    "aurora_hip_generation": (c_int64, [c_void_p]),
    "aurora_hip_output_vars": (c_int, [c_void_p, _PS, c_int]),
    "aurora_hip_set_time_ex": (c_int, [c_void_p, _PD, ctypes.POINTER(c_int32), c_int, c_void_p]),
    "aurora_hip_set_band": (c_int, [c_void_p, ctypes.POINTER(HipBand)]),
    "aurora_hip_band_rows": (c_int, [c_void_p, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]),
    "aurora_hip_band_staging_bytes": (c_int64, [c_void_p]),
    "aurora_hip_set_band_staging": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "aurora_hip_pos_scale_encoding": (c_int, [_PD, _PD, c_int, c_int, c_int, c_int, _PF, _PF]),
    "aurora_hip_band_partition": (c_int, [c_int, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32), c_int, c_int, c_int,
                                          ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]),
    "aurora_hip_band_plan": (c_int, [ctypes.POINTER(c_int32), ctypes.POINTER(c_int32), c_int, c_int, c_int,
                                     ctypes.POINTER(c_int32), ctypes.POINTER(HipPlanInfo), c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
})
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
