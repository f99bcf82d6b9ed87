"""Per-operator parity of libaurora_hip (through the ctypes C-ABI shim) against CPU references.

fp32 kernels are held to fp32-roundoff tolerances against fp64 torch-CPU evaluations; bf16
kernels are compared, on bf16-rounded inputs, against fp64 evaluations of the same rounded
inputs (so only accumulation order and the output rounding differ).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import aurora_oracle as oracle

pytestmark = pytest.mark.gpu

DEV = "cuda"


def lib():
    from aurora_amd.engine import lib as L

    L.load()
    return L


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 80, 160), (128, 128, 32), (1, 512, 64), (1000, 100, 96),
                                   (257, 1536, 512), (64, 27, 1408),
                                   # >= 1024 rows and N % 256 == 0: the 256 x 256 ring kernel
                                   (1030, 256, 32), (2050, 512, 160), (1279, 768, 64)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fp32(M, N, K, act):
    L = lib()
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a @ w.T + b
    ref = F.gelu(ref) if act == 1 else (F.silu(ref) if act == 2 else ref)
    ref = ref + r
    ld = (N + 3) // 4 * 4
    out = torch.full((M, ld), float("nan"), device=DEV)
    out2 = torch.zeros((M, ld), dtype=torch.bfloat16, device=DEV)
    L.linear(a.float().to(DEV), w.float().to(DEV), b.float().to(DEV), out, out2=out2,
             residual=r.float().to(DEV), act=act, n=N)
    torch.cuda.synchronize()
    assert relerr(out[:, :N], ref) < 2e-6
    assert relerr(out2[:, :N].float(), ref) < 5e-3
    if ld > N:
        assert torch.isnan(out[:, N:]).all()  # padding columns untouched


@pytest.mark.parametrize("M,N,K", [(1030, 256, 32), (2050, 512, 160), (1500, 768, 512), (4099, 1024, 2048)])
def test_linear_fp32_by_bf16_splitting_is_fp32_grade(M, N, K):
    """Large fp32 linears run as six bf16 MFMAs over an exact 3-way operand split; their error against an
    fp64 product must be of the order of the native fp32 MFMA kernel's (both are measured here, relative to
    the natural scale sum_k |a||w|), on operands spanning six decades."""
    L = lib()
    g = torch.Generator().manual_seed(11)
    mag = lambda *sh: 10.0 ** (torch.rand(*sh, generator=g, dtype=torch.float64) * 6 - 3)  # noqa: E731
    a = (rnd(M, K, seed=1) * mag(M, K)).float()
    w = (rnd(N, K, seed=2) * mag(N, K)).float()
    ref = a.double() @ w.double().T
    scale = a.double().abs() @ w.double().abs().T
    err = {}
    for mode in (0, 1, 2):
        out = torch.empty((M, N), device=DEV)
        with L.f32_gemm(mode):
            L.linear(a.to(DEV), w.to(DEV), None, out)
        torch.cuda.synchronize()
        err[mode] = ((out.double().cpu() - ref).abs() / scale).max().item()
    print(f"fp32 linear {M}x{N}x{K}: native MFMA err {err[0]:.2e}, 3xbf16 split err {err[1]:.2e}, "
          f"2xfp16 split err {err[2]:.2e}")
    assert err[0] < 2e-6, err
    assert err[1] < 2e-6 and err[1] < 2 * err[0] + 1e-7, err
    # two fp16 terms: as accurate, except that activations below 0.25 carry an ABSOLUTE error of up to 3e-8 (fp16
    # subnormal remainders) -- visible here only for K = 32, where a few tiny elements can dominate the sum
    assert err[2] < (4e-6 if K < 64 else 2e-6) and (K < 64 or err[2] < 2 * err[0] + 1e-7), err


@pytest.mark.parametrize("M,N,K", [(1030, 256, 96), (1279, 512, 128), (2050, 768, 160), (33000, 512, 512),
                                   (5000, 1024, 2048)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fp32_two_fp16_terms_full_epilogue(M, N, K, act):
    """The ping-pong two-term kernel (128 x 256 tiles, K-stages of 32; K = 96 is its three-stage minimum) with everything
    the epilogue can do: bias, activation, residual, second bf16 output, ragged last m-tile, padded leading dimension."""
    L = lib()
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a @ w.T + b
    ref = F.gelu(ref) if act == 1 else (F.silu(ref) if act == 2 else ref)
    ref = ref + r
    ld = N + 4
    out = torch.full((M, ld), float("nan"), device=DEV)
    out2 = torch.zeros((M, ld), dtype=torch.bfloat16, device=DEV)
    with L.f32_gemm(2):
        L.linear(a.float().to(DEV), w.float().to(DEV), b.float().to(DEV), out, out2=out2,
                 residual=r.float().to(DEV), act=act, n=N)
    torch.cuda.synchronize()
    assert relerr(out[:, :N], ref) < 2e-6
    assert relerr(out2[:, :N].float(), ref) < 5e-3
    assert torch.isnan(out[:, N:]).all()


@pytest.mark.parametrize("M,K", [(1000, 512), (5000, 2048), (128, 128), (66001, 512), (65, 512)])   # whole tiles + a ragged one, whole only, ragged only
@pytest.mark.parametrize("in_place", [True, False])
def test_linear_layernorm_fused_equals_the_two_kernels(M, K, in_place):
    """bf16 linear + AdaLN + residual in one launch (D = 512) against linear -> layernorm: the same rounded linear result
    enters the statistics, so the two differ by summation order only."""
    L = lib()
    N = 512
    a = rnd(M, K, seed=1).bfloat16().to(DEV)
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().to(DEV)
    b, gain, shift = (rnd(N, seed=s_).float().to(DEV) for s_ in (3, 4, 5))
    x = (rnd(M, N, seed=6) * 2).float().to(DEV)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    L.linear(a, w, b, y)
    x_ref, xb_ref = torch.empty_like(x), torch.empty_like(y)
    L.layernorm(y, gain, shift, res=x, out_f32=x_ref, out_t=xb_ref)
    x_in = x.clone()
    x_out = x_in if in_place else torch.full((M, N + 8), float("nan"), device=DEV)[:, :N]
    xb = torch.empty_like(y) if in_place else None
    L.linear_layernorm(a, w, b, gain, shift, x_in, x_out, xb)
    torch.cuda.synchronize()
    assert (x_out - x_ref).abs().max().item() < 2e-5
    if xb is not None:
        d = (xb.float() - xb_ref.float()).abs()
        assert d.max().item() <= 0.0625 and (d > 0).float().mean().item() < 1e-3   # a bf16 ulp at a rounding boundary
    else:
        assert torch.equal(x_in, x)   # the input stream is untouched
    with pytest.raises(ValueError):
        L.linear_layernorm(a, w[:256], b, gain, shift, x_in, x_in, None)   # only N = 512 rows fit one workgroup


def _unsplit(t):
    """fp16-pair layout -> (high halves, remainders) as fp32 tensors of the logical shape."""
    M, K = t.shape
    h16 = t.contiguous().view(torch.float16).view(M, K // 32, 2, 32)
    return h16[:, :, 0].reshape(M, K).float(), h16[:, :, 1].reshape(M, K).float()


def test_split_f16_layout_and_values():
    L = lib()
    x = (rnd(70, 96, seed=5) * 3).float().to(DEV)
    hi, lo = _unsplit(L.split_f16(x))
    ref_hi = x.half().float()
    assert torch.equal(hi, ref_hi) and torch.equal(lo, (x - ref_hi).half().float())
    hi, lo = _unsplit(L.split_f16(x, scale=64.0))
    assert torch.equal(hi, (x * 64).half().float()) and torch.equal(lo, (x * 64 - hi).half().float())
    with pytest.raises(ValueError):
        L.split_f16(x[:, :48].contiguous())   # K % 32 != 0


@pytest.mark.parametrize("M,N,K", [(1030, 256, 96), (2050, 768, 160), (5000, 1024, 2048)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fp32_presplit_operands_are_bit_identical(M, N, K, act):
    """A split is the same arithmetic wherever it happens: weights split at pack time, activations split by their
    producer and results written in the pair layout must reproduce the in-kernel split bit for bit."""
    L = lib()
    a = rnd(M, K, seed=1).float().to(DEV)
    w = rnd(N, K, seed=2, scale=K ** -0.5).float().to(DEV)
    b, r = rnd(N, seed=3).float().to(DEV), rnd(M, N, seed=4).float().to(DEV)
    base = torch.empty((M, N), device=DEV)
    with L.f32_gemm(2):
        L.linear(a, w, b, base, residual=r, act=act)
    ws, a_s = L.split_f16(w, scale=64.0), L.split_f16(a)
    for flags, a_in in ((L.F32_W_SPLIT, a), (L.F32_W_SPLIT | L.F32_A_SPLIT, a_s)):
        out = torch.full((M, N), float("nan"), device=DEV)
        L.linear(a_in, ws, b, out, residual=r, act=act, presplit=flags)
        assert torch.equal(out, base), flags
    out = torch.zeros((M, N), device=DEV)
    L.linear(a_s, ws, b, out, residual=r, act=act, presplit=L.F32_W_SPLIT | L.F32_A_SPLIT | L.F32_C_SPLIT)
    torch.cuda.synchronize()
    assert torch.equal(out, L.split_f16(base))
    # contract violations are refused, not mis-executed
    with pytest.raises(ValueError):
        L.linear(a_s, w, b, out, presplit=L.F32_A_SPLIT)             # pre-split activations need pre-split weights


@pytest.mark.parametrize("M,N,K", [(1030, 128, 96), (5000, 384, 1024), (70000, 128, 1024)])
def test_linear_fp32_narrow_two_term_tiles_equal_the_wide_ones(M, N, K):
    """The 256 x 128 tiling of the two-term kernel (N a multiple of 128 only: the decoder's output heads, 80 real columns)
    multiplies in the same K order as the 128 x 256 one: bit-identical to it on weights zero-padded to 256 columns."""
    L = lib()
    a = rnd(M, K, seed=1).float().to(DEV)
    w = rnd(N, K, seed=2, scale=K ** -0.5).float().to(DEV)
    b = rnd(N, seed=3).float().to(DEV)
    Np = (N + 255) // 256 * 256
    wp, bp = torch.zeros((Np, K), device=DEV), torch.zeros(Np, device=DEV)
    wp[:N], bp[:N] = w, b
    ws, wps, a_s = L.split_f16(w, scale=64.0), L.split_f16(wp, scale=64.0), L.split_f16(a)
    wide = torch.empty((M, Np), device=DEV)
    L.linear(a_s, wps, bp, wide, presplit=L.F32_W_SPLIT | L.F32_A_SPLIT)
    for flags, a_in in ((L.F32_W_SPLIT, a), (L.F32_W_SPLIT | L.F32_A_SPLIT, a_s)):
        out = torch.full((M, N), float("nan"), device=DEV)
        L.linear(a_in, ws, b, out, presplit=flags)
        torch.cuda.synchronize()
        assert torch.equal(out, wide[:, :N]), flags
    ref = a.double().cpu() @ w.double().cpu().T + b.double().cpu()
    assert relerr(out, ref) < 2e-6


@pytest.mark.parametrize("amax", [3.0, 1.0e6])
def test_linear_fp32_presplit_weights_guarded_pair(amax):
    """Pre-split weights with a guard: the two-term launch runs iff the guard holds, a mode-1 launch with the same guard
    iff it does not -- together they are the guarded mode-2 call."""
    L = lib()
    M, N, K = 2050, 512, 256
    a = (rnd(M, K, seed=21) * amax).float().to(DEV)
    w = rnd(N, K, seed=22, scale=K ** -0.5).float().to(DEV)
    ws = L.split_f16(w, scale=64.0)
    g = L.absmax(a)
    base = torch.empty((M, N), device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    with L.bounded_activations(guard=(g, 16384.0)):
        L.linear(a, w, None, base)
        L.linear(a, ws, None, out, presplit=L.F32_W_SPLIT)
    with L.f32_gemm(1, guard=(g, 16384.0)):
        L.linear(a, w, None, out)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, base)


@pytest.mark.parametrize("amax", [3.0, 1.0e6])
def test_linear_fp32_guarded_chain_switches_format_and_kernels_together(amax):
    """A guarded producer writes fp16 pairs iff the guard holds (its mode-1 twin fp32 otherwise) and the guarded consumer
    reads the same buffer accordingly: the chain equals two guarded mode-2 linears on fp32 buffers, whatever the guard."""
    L = lib()
    M, K, D, N = 2050, 160, 512, 1024
    a = (rnd(M, K, seed=31) * amax).float().to(DEV)
    w1 = rnd(D, K, seed=32, scale=K ** -0.5).float().to(DEV)
    w2 = rnd(N, D, seed=33, scale=D ** -0.5).float().to(DEV)
    b1 = rnd(D, seed=34).float().to(DEV)
    g = L.absmax(a)
    limit = 1000.0
    mid_ref, out_ref = torch.empty((M, D), device=DEV), torch.empty((M, N), device=DEV)
    with L.bounded_activations(guard=(g, limit)):
        L.linear(a, w1, b1, mid_ref)
        L.linear(mid_ref, w2, None, out_ref)
    w1s, w2s = L.split_f16(w1, scale=64.0), L.split_f16(w2, scale=64.0)
    mid, out = torch.empty((M, D), device=DEV), torch.full((M, N), float("nan"), device=DEV)
    with L.f32_gemm(2, guard=(g, limit)):
        L.linear(a, w1s, b1, mid, presplit=L.F32_W_SPLIT | L.F32_C_SPLIT)            # pairs iff g < limit
        L.linear(mid, w2s, None, out, presplit=L.F32_W_SPLIT | L.F32_A_SPLIT)
    with L.f32_gemm(1, guard=(g, limit)):
        L.linear(a, w1, b1, mid)                                                      # fp32 iff not
        L.linear(mid, w2, None, out)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, out_ref)
    assert torch.equal(mid, L.split_f16(mid_ref) if amax < limit else mid_ref)


@pytest.mark.parametrize("holds", [True, False])
def test_perceiver_attention_pair_output_follows_the_guard(holds):
    L = lib()
    B, cols, Lq, Lk, heads, hd = 1, 500, 13, 3, 16, 64
    inner = heads * hd
    q = rnd(Lq, inner, seed=1).float().to(DEV)
    kv = rnd(Lk * cols, 2 * inner, seed=2).float().to(DEV)
    ref = L.perceiver_attention(q, 0, kv, torch.empty(cols * Lq, inner, device=DEV), B, cols, cols * Lk, cols, Lq, Lk, heads, hd)
    word = torch.tensor([1.0 if holds else 3.0], device=DEV)
    out = L.perceiver_attention(q, 0, kv, torch.empty(cols * Lq, inner, device=DEV), B, cols, cols * Lk, cols, Lq, Lk, heads, hd,
                                pair_guard=(word, 2.0))
    torch.cuda.synchronize()
    assert torch.equal(out, L.split_f16(ref) if holds else ref)


def test_layernorm_split_output():
    L = lib()
    M, D = 1000, 1024
    y, res = rnd(M, D, seed=1).float().to(DEV) * 3, rnd(M, D, seed=2).float().to(DEV)
    gw, gb = rnd(D, seed=3).float().to(DEV), rnd(D, seed=4).float().to(DEV)
    plain, both = torch.empty_like(y), torch.empty_like(y)
    sp = torch.empty_like(y)
    L.layernorm(y, gw, gb, res=res, out_f32=plain)
    L.layernorm(y, gw, gb, res=res, out_f32=both, out_t=sp, split_t=True)
    torch.cuda.synchronize()
    assert torch.equal(plain, both) and torch.equal(sp, L.split_f16(plain))
    only = torch.empty_like(y)
    L.layernorm(y, gw, gb, res=res, out_t=only, split_t=True)          # the fp32 copy is optional
    assert torch.equal(only, sp)
    # a residual in the pair layout is worth high half + remainder: the fp32 value to 2^-23 relative
    from_pairs = torch.empty_like(y)
    L.layernorm(y, gw, gb, res=L.split_f16(res), out_f32=from_pairs, split_res=True)
    hi, lo = _unsplit(L.split_f16(res))
    exact = torch.empty_like(y)
    L.layernorm(y, gw, gb, res=hi + lo, out_f32=exact)
    torch.cuda.synchronize()
    assert torch.equal(from_pairs, exact)
    assert (from_pairs - plain).abs().max().item() <= 2e-6   # |res| <= 4.5: 2^-23 of it, plus one rounding of the sum


@pytest.mark.parametrize("amax", [3.0, 1.0e6])
def test_linear_fp32_range_guard_picks_the_split_on_the_device(amax):
    """With a guard, mode 2 launches both operand splits and the device word decides: activations inside fp16's range
    take the two-term fp16 split, anything larger (here 1e6: fp16 would overflow to inf) the three-term bf16 one."""
    L = lib()
    M, N, K = 2050, 512, 256
    a = (rnd(M, K, seed=21) * amax).float()
    w = rnd(N, K, seed=22, scale=K ** -0.5).float()
    ref = a.double() @ w.double().T
    a_d = a.to(DEV)
    out = torch.empty((M, N), device=DEV)
    measured = L.absmax(a_d)
    with L.bounded_activations(guard=(measured, 16384.0)):
        L.linear(a_d, w.to(DEV), None, out)
    torch.cuda.synchronize()
    assert abs(measured.item() - a.abs().max().item()) == 0.0
    assert torch.isfinite(out).all()
    assert relerr(out, ref) < 2e-6
    assert L._f32_state() == (-1, None, 0.0)   # the per-thread mode is restored; the library has no state at all


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (130, 64, 64), (1000, 1536, 512), (129, 80, 2048),
                                   # the 256 x 256 ring kernel: 2, 4, 6 and 32 stages, ragged M
                                   (1030, 256, 64), (1500, 512, 128), (1100, 768, 192), (4099, 1536, 1024),
                                   # more tiles than CUs, ragged last m-tile: the persistent ring kernel (N >= 1024) and the
                                   # plain one (N = 512)
                                   (66001, 512, 256), (33000, 1024, 512),
                                   # per-rank shapes of an 8-way latitude-band split: the cost model picks 256 x 128 tiles
                                   # (two workgroups per CU) for the first two, 128 x 128 tiles for the third
                                   (34560, 512, 512), (8100, 3072, 1024), (2160, 2048, 2048)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_bf16(M, N, K, act):
    L = lib()
    a = rnd(M, K, seed=1).bfloat16()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16()
    b, r = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.double() @ w.double().T + b
    ref = F.gelu(ref) if act == 1 else ref
    ref = ref + r
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    out2 = torch.zeros((M, N), device=DEV)
    L.linear(a.to(DEV), w.to(DEV), b.float().to(DEV), out, out2=out2, residual=r.float().to(DEV), act=act)
    torch.cuda.synchronize()
    # fp32 copy: exact products, fp32 accumulation; the bf16 kernel's GELU uses a 4e-7-accurate erf
    assert relerr(out2, ref) < (5e-6 if act == 0 else 2e-5)
    assert relerr(out.float(), ref) < 5e-3    # bf16 copy: one rounding


@pytest.mark.parametrize("M,N,K,split", [(2160, 2048, 8192, 0), (2160, 2048, 8192, 2), (2160, 2048, 8192, 3), (2160, 2048, 8192, 5),
                                           (1100, 512, 4096, 4), (4320, 1024, 4096, 2)])
@pytest.mark.parametrize("epilogue", ["plain", "dual"])
def test_linear_ws_split_k_equals_the_unsplit_product(M, N, K, split, epilogue):
    """Split-K inside one launch (aurora_hip_linear_ws: a latitude band's few-tile / long-K linears): the fp32 copy of the
    result equals the fp64 product like the un-split kernel's, the bf16 copy differs from the un-split kernel's by at most
    one rounding (the slices are added in fp32 before the one rounding to bf16), the result does not depend on which
    workgroup came last (two runs: bit-identical), and the tickets are left zero for the next launch."""
    L = lib()
    a = rnd(M, K, seed=1).bfloat16()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16()
    b = rnd(N, seed=3)
    ref = a.double() @ w.double().T + b
    ad, wd, bd = a.to(DEV), w.to(DEV), b.float().to(DEV)
    want = L.linear_workspace(M, N, K)
    assert want > 0 or split != 0 or M * N > 2160 * 2048      # the library wants to split the band shape by itself
    ws = torch.empty(max(want, 8 * ((M + 255) // 256) * (N // 256) * 256 * 256 * 4), dtype=torch.uint8, device=DEV)
    tickets = torch.zeros(4096, dtype=torch.int32, device=DEV)
    outs = []
    for _ in range(2):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        out2 = torch.full((M, N), float("nan"), device=DEV) if epilogue == "dual" else None
        L.linear_ws(ad, wd, bd, out, ws, tickets, split=split, out2=out2)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        outs.append((out, out2))
    assert torch.equal(outs[0][0], outs[1][0])
    if epilogue == "dual":
        assert torch.equal(outs[0][1], outs[1][1])
        assert relerr(outs[0][1], ref) < 5e-6
    assert relerr(outs[0][0].float(), ref) < 5e-3
    plain = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    L.linear(ad, wd, bd, plain)
    torch.cuda.synchronize()
    d = (outs[0][0].float() - plain.float()).abs()
    assert (d <= plain.float().abs() * 2.0 ** -7 + 1e-6).all()     # at most one bf16 ulp apart


def test_linear_broadcast_residual_row_and_strided_views():
    L = lib()
    M, N, K = 200, 96, 64
    a_full, w, row = rnd(M, 2 * K, seed=5), rnd(N, K, seed=6), rnd(1, N, seed=7)
    a_dev = a_full.float().to(DEV)
    out_full = torch.zeros((M, 3 * N), device=DEV)
    L.linear(a_dev[:, K:], w.float().to(DEV), None, out_full[:, N:2 * N],
             residual=row.float().to(DEV).expand(M, N))
    torch.cuda.synchronize()
    ref = a_full[:, K:] @ w.T + row
    assert relerr(out_full[:, N:2 * N], ref) < 2e-6
    assert (out_full[:, :N] == 0).all() and (out_full[:, 2 * N:] == 0).all()


def test_linear_rejects_bad_k():
    L = lib()
    a, w = torch.zeros((4, 40), device=DEV), torch.zeros((8, 40), device=DEV)
    with pytest.raises(ValueError, match="multiple"):
        L.linear(a, w, None, torch.zeros((4, 8), device=DEV))


# ------------------------------------------------------------------------------------------
# window attention
# ------------------------------------------------------------------------------------------
def attention_reference(qkv, bias, tok, grp, B, Ltok, D, heads, scaled_mask=False):
    """Gather through the (already reference-checked) tables, SDPA per window, scatter back.  `scaled_mask`: the WRONG mask
    -100 x (group difference) of an earlier build of the bf16 kernel, for the test that tells the two apart."""
    nW, N = tok.shape
    hd = D // heads
    out = torch.zeros((B, Ltok, D), dtype=torch.float64)
    tok_t = torch.from_numpy(tok.astype(np.int64))
    for b in range(B):
        rows = torch.where(tok_t[..., None] >= 0, qkv[b][tok_t.clamp(min=0)], bias.expand(nW, N, 3 * D))
        q, k, v = rows.reshape(nW, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        mask = None
        if grp is not None:
            g = torch.from_numpy(grp.astype(np.int64))
            mask = torch.where(g[:, None, :] != g[:, :, None], -100.0, 0.0)[:, None].double()
            if scaled_mask:
                mask = (-100.0 * (g[:, None, :] ^ g[:, :, None]))[:, None].double()
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        o = o.transpose(1, 2).reshape(nW, N, D)
        valid = tok_t >= 0
        out[b][tok_t[valid]] = o[valid]
    return out


ATTN_CASES = [((4, 12, 24), (2, 6, 12), 2), ((4, 13, 26), (2, 6, 12), 1), ((4, 7, 13), (2, 6, 12), 2),
              ((4, 4, 8), (2, 6, 12), 4), ((4, 1, 2), (2, 6, 12), 1), ((2, 5, 9), (2, 3, 4), 1)]


@pytest.mark.parametrize("res,window,heads", ATTN_CASES)
@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention(res, window, heads, shifted, dtype):
    from aurora_amd.engine import geometry

    L = lib()
    B, D = 2, 64 * heads
    Ltok = res[0] * res[1] * res[2]
    tok, grp, _ = geometry.window_tables(res, window, shifted)
    qkv = rnd(B, Ltok, 3 * D, seed=11, scale=2.0).to(dtype)
    bias = rnd(3 * D, seed=12)
    bias_used = bias.to(dtype).double() if dtype == torch.bfloat16 else bias.float().double()
    ref = attention_reference(qkv.double(), bias_used, tok, grp, B, Ltok, D, heads)
    out = torch.full((B, Ltok, D), 7.0, dtype=dtype, device=DEV)
    L.window_attention(qkv.to(DEV).contiguous(), bias.float().to(DEV), out,
                       torch.from_numpy(tok).to(DEV), None if grp is None else torch.from_numpy(grp).to(DEV),
                       B, Ltok, D, heads)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(out.float(), ref) < tol


@pytest.mark.parametrize("res,heads", [((4, 12, 24), 2), ((4, 13, 26), 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_window_attention_mask_is_the_references_literal_minus_100(res, heads, dtype):
    """The shift mask is 0 / -100, NOT -inf (swin3d.py:357-358): a key of another group whose raw score
    beats the query's own group by MORE than 100 still wins the softmax.  Adversarial inputs: the keys of one mask group
    score 128 against every query (q . k / 8 with q_0 = 16, k_0 = 64), every other key about 0 -- queries of the other
    groups then see those keys at 128 - 100 = 28 above their own group.  A mask of -100 x (group difference), or -inf, puts
    the weight on the wrong keys; the reference is fp64 SDPA with the literal -100."""
    from aurora_amd.engine import geometry

    L = lib()
    window, B, D = (2, 6, 12), 1, 64 * heads
    Ltok = res[0] * res[1] * res[2]
    tok, grp, _ = geometry.window_tables(res, window, True)
    assert grp is not None
    qkv = rnd(B, Ltok, 3, heads, 64, seed=31, scale=0.5)
    # the loud keys: the tokens of the smallest group id != the first token's group in a window that mixes groups with a
    # group difference >= 2 somewhere (where -100 x difference and -100 part ways)
    mixed = [w for w in range(tok.shape[0]) if len(set(grp[w][tok[w] >= 0].tolist())) >= 2]
    assert mixed and any((int(a) ^ int(b)) >= 2 for w in mixed for a in set(grp[w].tolist()) for b in set(grp[w].tolist()))
    loud = np.zeros(Ltok, dtype=bool)
    for w in mixed:
        ids = sorted(set(grp[w][tok[w] >= 0].tolist()))
        loud[tok[w][(grp[w] == ids[-1]) & (tok[w] >= 0)]] = True
    qkv[:, :, 0, :, 0] = 16.0
    qkv[:, :, 1, :, 0] = torch.where(torch.from_numpy(loud)[None, :, None], 64.0, 0.0).to(qkv.dtype)
    qkv = qkv.reshape(B, Ltok, 3 * D).to(dtype)
    bias = torch.zeros(3 * D)
    ref = attention_reference(qkv.double(), bias.double(), tok, grp, B, Ltok, D, heads)
    out = torch.full((B, Ltok, D), 7.0, dtype=dtype, device=DEV)
    L.window_attention(qkv.to(DEV).contiguous(), bias.float().to(DEV), out, torch.from_numpy(tok).to(DEV),
                       torch.from_numpy(grp).to(DEV), B, Ltok, D, heads)
    torch.cuda.synchronize()
    # (fp32: a score of 128 carries an absolute rounding error of ~128 x 2^-23, which the exponential turns into a relative
    # error of the weights -- ten times the tolerance of the ordinary cases, four orders below what the wrong mask does)
    err = relerr(out.float(), ref)
    assert err < (2e-4 if dtype == torch.float32 else 1.5e-2), err
    # the check has teeth: -100 x (group difference) gives something else on these inputs
    wrong = attention_reference(qkv.double(), bias.double(), tok, grp, B, Ltok, D, heads, scaled_mask=True)
    assert relerr(wrong, ref) > 0.1


@pytest.mark.parametrize("M,heads,K", [(300, 2, 128), (1300, 8, 512), (70000, 8, 512), (2160, 32, 2048), (257, 4, 64)])
def test_linear_planes_and_attention_on_planes_equal_the_row_layout(M, heads, K):
    """q | k | v of a token side by side, one attention head per plane (aurora_hip_linear_planes ->
    aurora_hip_window_attention_planes): every kernel
    the dispatcher may pick for these shapes (128 x 128, 256 x 128, 256 x 256 tiles) writes the same bits to the same
    (row, column) as the row layout, planes longer than M leave their tail alone, and the attention reads them to the same
    output bit for bit."""
    from aurora_amd.engine import geometry

    L = lib()
    D = 64 * heads
    a = rnd(M, K, seed=1).bfloat16().to(DEV)
    w = rnd(3 * D, K, seed=2, scale=K ** -0.5).bfloat16().to(DEV)
    b = rnd(3 * D, seed=3).float().to(DEV)
    rows = torch.empty((M, 3 * D), dtype=torch.bfloat16, device=DEV)
    L.linear(a, w, b, rows)
    pad = 5
    planes = torch.full((heads, M + pad, 3, 64), 9.0, dtype=torch.bfloat16, device=DEV)
    L.linear_planes(a, w, b, planes)
    torch.cuda.synchronize()
    as_rows = lambda pl: pl.permute(1, 2, 0, 3).reshape(pl.shape[1], -1)   # (rows, sel, head, 64) -> (rows, sel * D)  # noqa: E731
    assert torch.equal(as_rows(planes[:, :M]), rows)
    assert (planes[:, M:] == 9.0).all()
    # a latitude band's halo projection: k | v only, written into rows [M - 7, M) behind the q part of every row
    kv = torch.full_like(planes, 3.0)
    L.linear_planes(a[:7], w[D:], b[D:], kv[:, M - 7:], sel0=1)   # (a row range: the plane stride is the whole buffer's)
    torch.cuda.synchronize()
    assert torch.equal(as_rows(kv[:, M - 7:M, 1:]), rows[:7, D:])
    assert (kv[:, :, 0] == 3.0).all() and (kv[:, :M - 7] == 3.0).all() and (kv[:, M:] == 3.0).all()
    # attention over windows of the first tokens (one batch element; the planes keep their padding rows)
    res = (2, 6, 12) if M >= 144 else None
    if res is None:
        return
    n_tok_grid = min(M // 144, 40) * 144
    grid = (2, 6, 12 * (n_tok_grid // 144))
    tok, grp, _ = geometry.window_tables(grid, (2, 6, 12), True)
    tok_d, grp_d = torch.from_numpy(tok).to(DEV), None if grp is None else torch.from_numpy(grp).to(DEV)
    Lt = n_tok_grid
    out_rows = torch.zeros((Lt, D), dtype=torch.bfloat16, device=DEV)
    out_planes = torch.ones_like(out_rows)
    L.window_attention(rows[:Lt].contiguous(), b, out_rows, tok_d, grp_d, 1, Lt, D, heads)
    L.window_attention(planes[:, :Lt].contiguous(), b, out_planes, tok_d, grp_d, 1, Lt, D, heads, planes=True)
    torch.cuda.synchronize()
    assert torch.equal(out_rows, out_planes)


# ------------------------------------------------------------------------------------------
# layer norms
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [64, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_residual(D, dtype):
    L = lib()
    M = 37
    y = (rnd(M, D, seed=1, scale=3.0) + 0.5).to(dtype)
    gain, shift, res = rnd(D, seed=2) + 1, rnd(D, seed=3), rnd(5, D, seed=4)
    ref = F.layer_norm(y.double(), (D,), eps=1e-5) * gain + shift + res[torch.arange(M) % 5]
    out_f = torch.zeros((M, D), device=DEV)
    out_t = torch.zeros((M, D), dtype=dtype, device=DEV)
    L.layernorm(y.to(DEV), gain.float().to(DEV), shift.float().to(DEV), res=res.float().to(DEV), res_mod=5,
                out_f32=out_f, out_t=out_t)
    torch.cuda.synchronize()
    assert relerr(out_f, ref) < 3e-6
    assert relerr(out_t.float(), ref) < (3e-6 if dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("D", [256, 512, 1024, 2048])
@pytest.mark.parametrize("M", [12001, 70003])
def test_layernorm_many_rows(D, M):
    """bf16 launches with tens of thousands of rows, as the backbone issues them (the small-M cases above take the variant
    that prefetches the residual row): every row written exactly once, ragged row counts, with and without gain / shift /
    residual / either output, in place on the residual stream.  (Round 5 also ran this against a persistent, grid-stride
    form of the kernel, built to co-reside with another stream's GEMM tiles and deleted again: profiles/README.md.)"""
    L = lib()
    y = (rnd(M, D, seed=1, scale=3.0) + 0.5).bfloat16()
    gain, shift, res = rnd(D, seed=2) + 1, rnd(D, seed=3), rnd(M, D, seed=4)
    ref = F.layer_norm(y.double(), (D,), eps=1e-5) * gain + shift + res
    x = res.float().to(DEV)                       # x <- x + LN(y) gain + shift, in place, plus the bf16 shadow
    shadow = torch.zeros((M, D), dtype=torch.bfloat16, device=DEV)
    L.layernorm(y.to(DEV), gain.float().to(DEV), shift.float().to(DEV), res=x, out_f32=x, out_t=shadow)
    torch.cuda.synchronize()
    assert relerr(x, ref) < 3e-6 and relerr(shadow.float(), ref) < 5e-3
    assert torch.equal(shadow, x.bfloat16())
    plain = torch.full((M, D), 9.0, dtype=torch.bfloat16, device=DEV)   # no gain / shift / residual, bf16 output only
    L.layernorm(y.to(DEV), None, None, out_t=plain)
    torch.cuda.synchronize()
    assert relerr(plain.float(), F.layer_norm(y.double(), (D,), eps=1e-5)) < 5e-3


def test_layernorm_in_place_on_column_block():
    L = lib()
    M, inner = 50, 128
    kv = rnd(M, 2 * inner, seed=9).float().to(DEV)
    orig = kv.clone()
    w, b = rnd(inner, seed=1) + 1, rnd(inner, seed=2)
    L.layernorm(kv, w.float().to(DEV), b.float().to(DEV), out_f32=kv, d=inner)
    torch.cuda.synchronize()
    ref = F.layer_norm(orig[:, :inner].double().cpu(), (inner,), w, b)
    assert relerr(kv[:, :inner], ref) < 3e-6
    assert torch.equal(kv[:, inner:], orig[:, inner:])


@pytest.mark.parametrize("H,W", [(12, 24), (13, 26), (7, 13)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_merge_and_split(H, W, dtype):
    L = lib()
    B, C, D = 2, 4, 64
    x = rnd(B, C * H * W, D, seed=1)
    sd = {"m.norm.weight": rnd(4 * D, seed=2) + 1, "m.norm.bias": rnd(4 * D, seed=3),
          "m.reduction.weight": torch.eye(4 * D, dtype=torch.float64)}
    ref = oracle.patch_merge(sd, "m", x, (C, H, W))  # identity reduction: the LN output itself
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    out = torch.zeros((B * C * H2 * W2, 4 * D), dtype=dtype, device=DEV)
    L.merge_ln(x.float().to(DEV).contiguous(), sd["m.norm.weight"].float().to(DEV),
               sd["m.norm.bias"].float().to(DEV), out, B, C, H, W, D)
    torch.cuda.synchronize()
    tol = 3e-6 if dtype == torch.float32 else 5e-3
    assert relerr(out.float(), ref.reshape(-1, 4 * D)) < tol

    # split: (C, H2, W2) with 2D' features back to (C, H, W), crop = odd remainders
    Dq = 32
    y = rnd(B, C * H2 * W2, 4 * Dq, seed=5).to(dtype)
    sd2 = {"s.lin1.weight": torch.eye(4 * Dq, dtype=torch.float64), "s.norm.weight": rnd(Dq, seed=6) + 1,
           "s.norm.bias": rnd(Dq, seed=7), "s.lin2.weight": torch.eye(Dq, dtype=torch.float64)}
    ref2 = oracle.patch_split(sd2, "s", y.double(), (C, H2, W2), (0, H % 2, W % 2))
    out2 = torch.zeros((B * C * H * W, Dq), dtype=dtype, device=DEV)
    L.split_ln(y.to(DEV).contiguous(), sd2["s.norm.weight"].float().to(DEV), sd2["s.norm.bias"].float().to(DEV),
               out2, B, C, H2, W2, Dq, H % 2, W % 2)
    torch.cuda.synchronize()
    assert relerr(out2.float(), ref2.reshape(-1, Dq)) < tol


# ------------------------------------------------------------------------------------------
# embed / unembed
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,Wp", [(3, 7), (4, 7), (10, 7), (3, 37), (10, 32), (4, 37)])   # 37: two full spans of the any-P kernel and a ragged one
def test_patchify_matches_conv_unfold(P, Wp):
    L = lib()
    B, T, C, Hp, V = 2, 2, 3, 5, 3
    H, W = Hp * P, Wp * P
    x = rnd(B, T, V, C, H + 1, W, seed=1, scale=5.0).float()  # one extra latitude row (cropped view)
    static = rnd(H, W, seed=2).float()
    loc, sc = rnd(V + 1, C, seed=3).float(), (rnd(V + 1, C, seed=4).abs() + 0.5).float()
    xd, sdv = x.to(DEV), static.to(DEV)
    descs, keep = [], []
    for v in range(V + 1):
        lo, inv = loc[v].to(DEV).contiguous(), (1.0 / sc[v].double()).float().to(DEV).contiguous()
        keep += [lo, inv]
        if v < V:
            t = xd[:, :, v, :, :H, :]
            sb, st, scs, sh, sw = t.stride()
            descs.append(L.PatchVar(t.data_ptr(), sb, st, scs, sh, sw, lo.data_ptr(), inv.data_ptr(),
                                    1 if v == 1 else 0, 0.0, 0.0, 0.0))
        else:
            descs.append(L.PatchVar(sdv.data_ptr(), 0, 0, 0, W, 1, lo.data_ptr(), inv.data_ptr(), 0, 0.0, 0.0, 0.0))
    K = (V + 1) * T * P * P
    Kpad = (K + 31) // 32 * 32 + 32   # (always some K padding for the kernel to zero)
    out = torch.full((C * B * Hp * Wp, Kpad), float("nan"), device=DEV)
    word = torch.full((1,), 0.25, device=DEV)    # folded into, not overwritten
    L.patchify(descs, out, 0, K, B, T, C, Hp, Wp, P, absmax=word)
    torch.cuda.synchronize()
    assert float(word) == float(out[:, :K].abs().max())   # the guard word of the linears that read `out`
    # reference: normalise, clamp var 1, unfold
    xn = (x[..., :H, :] - loc[:V, :, None, None]) / sc[:V, :, None, None]
    xn[:, :, 1] = xn[:, :, 1].clamp(min=0)
    sn = ((static[None] - loc[V, :, None, None]) / sc[V, :, None, None])  # (C, H, W)
    full = torch.cat([xn, sn[None, None, None].expand(B, T, 1, C, H, W)], dim=2)  # (B,T,V+1,C,H,W)
    pat = full.reshape(B, T, V + 1, C, Hp, P, Wp, P).permute(3, 0, 4, 6, 2, 1, 5, 7)  # c b hp wp v t i j
    ref = pat.reshape(C * B * Hp * Wp, K)
    assert relerr(out[:, :K], ref) < 2e-6
    assert (out[:, K:] == 0).all()


@pytest.mark.parametrize("P,Wp", [(4, 6), (3, 6), (10, 6), (10, 29), (4, 67)])   # Wp * P past one 256-pixel block as well
def test_unpatchify_matches_oracle(P, Wp):
    L = lib()
    B, CA, Hp, V = 2, 3, 4, 3
    H, W = Hp * P, Wp * P
    y = rnd(B * Hp * Wp * CA, V * P * P, seed=1).float()
    loc, sc = rnd(V, CA, seed=2).float(), (rnd(V, CA, seed=3).abs() + 0.5).float()
    # oracle layout: (B, L, C, V*P*P) with V fastest
    y_or = y.reshape(B, Hp * Wp, CA, V, P * P).permute(0, 1, 2, 4, 3).reshape(B, Hp * Wp, CA, P * P * V)
    ref = oracle.unpatchify(y_or.double(), V, H, W, P)  # (B, V, C, H, W)
    ref[:, 1] = ref[:, 1].clamp(min=0)
    ref = ref * sc.double()[None, :, :, None, None] + loc.double()[None, :, :, None, None]
    out = torch.zeros((V, B, CA, H, W), device=DEV)
    keep, descs = [], []
    for v in range(V):
        lo, s_ = loc[v].to(DEV).contiguous(), sc[v].to(DEV).contiguous()
        keep += [lo, s_]
        descs.append(L.unpatch_var(out[v].data_ptr(), lo.data_ptr(), s_.data_ptr(), 1 if v == 1 else 0, v * P * P))
    L.unpatchify(y.to(DEV), descs, B, CA, Hp, Wp, P)
    torch.cuda.synchronize()
    assert relerr(out.permute(1, 0, 2, 3, 4), ref) < 2e-6


@pytest.mark.parametrize("Lq,Lk,hd", [(3, 13, 32), (13, 3, 64), (3, 4, 16), (13, 1, 64), (9, 4, 64), (13, 5, 64)])   # <= 4 keys: the few-keys form
def test_perceiver_attention(Lq, Lk, hd):
    L = lib()
    B, cols, heads = 2, 50, 4
    inner = heads * hd
    q = rnd(Lq, inner, seed=1)
    kv = rnd(B * Lk * cols, 2 * inner, seed=2)  # rows (b, j, l)
    out = torch.zeros((B * cols * Lq, inner), device=DEV)
    L.perceiver_attention(q.float().to(DEV), 0, kv.float().to(DEV), out, B, cols, Lk * cols, cols, Lq, Lk, heads, hd)
    torch.cuda.synchronize()
    kvr = kv.reshape(B, Lk, cols, 2, heads, hd).permute(3, 0, 2, 4, 1, 5)  # (2, B, cols, heads, Lk, hd)
    qq = q.reshape(Lq, heads, hd).permute(1, 0, 2)[None, None].expand(B, cols, -1, -1, -1)
    ref = F.scaled_dot_product_attention(qq, kvr[0], kvr[1])  # (B, cols, heads, Lq, hd)
    ref = ref.permute(0, 1, 3, 2, 4).reshape(B * cols * Lq, inner)
    assert relerr(out, ref) < 3e-6


def _reassoc_reference(q, kv, w, B, cols, Lq, Lk, heads, hd):
    """fp64: F.scaled_dot_product_attention over the Lk keys of every column, then F.linear with to_out's weight
    (perceiver.py:141-152) -- rows (b, col, l)."""
    inner = heads * hd
    kvr = kv.reshape(B, Lk, cols, 2, heads, hd).permute(3, 0, 2, 4, 1, 5)  # (2, B, cols, heads, Lk, hd)
    qq = q.reshape(Lq, heads, hd).permute(1, 0, 2)[None, None].expand(B, cols, -1, -1, -1)
    att = F.scaled_dot_product_attention(qq, kvr[0], kvr[1])  # (B, cols, heads, Lq, hd)
    return F.linear(att.permute(0, 1, 3, 2, 4).reshape(B * cols * Lq, inner), w)


# (the production decoder: 13 levels, 3 latent keys, 16 x 64, D = 1024; a ragged column count; the small level counts)
@pytest.mark.parametrize("Lq,heads,N,cols,B", [(13, 16, 1024, 300, 1), (13, 16, 1024, 77, 2), (4, 2, 128, 50, 1), (3, 4, 256, 33, 2)])
def test_perceiver_out_reassociated_equals_attention_then_to_out(Lq, heads, N, cols, B):
    L = lib()
    Lk, hd = 3, 64
    inner = heads * hd
    q = rnd(Lq, inner, seed=1)
    kv = rnd(B * Lk * cols, 2 * inner, seed=2)   # rows (b, j, col)
    w = rnd(N, inner, seed=3, scale=inner ** -0.5)
    ref = _reassoc_reference(q, kv, w, B, cols, Lq, Lk, heads, hd)
    assert L.load().aurora_hip_perceiver_out_supported(Lq, Lk, heads, hd, N) == 1
    P, Vp = L.perceiver_probs(q.float().to(DEV), kv.float().to(DEV), B, cols, Lk * cols, cols, Lq, Lk, heads, hd)
    w_pairs = L.split_f16(w.float().to(DEV), scale=64.0)
    out = torch.full((B * cols * Lq, N), float("nan"), device=DEV)
    L.perceiver_out(Vp, w_pairs, P, out, B * cols, Lq, Lk, heads, hd)
    torch.cuda.synchronize()
    # the weights of keys 0 and 1 (the third follows from their sum) -- level 0 as it is, the other levels as differences to
    # level 0 (how perceiver_out combines them) -- as 16 pairs per (column, head): pair j * NLP + lp = levels (2 lp, 2 lp + 1)
    # of key j, NLP = ceil(Lq / 2): the operands of the packed FMAs; zeros behind the last level
    kvr = kv.reshape(B, Lk, cols, 2, heads, hd)
    sc = torch.einsum("lhd,bjchd->bchlj", q.reshape(Lq, heads, hd), kvr[:, :, :, 0]) / 8.0
    p_ref = torch.softmax(sc, dim=-1).reshape(B * cols, heads, Lq, Lk)
    p_got = P.reshape(B * cols, heads, 64)[:, :, :32].cpu().double()
    nlp = (Lq + 1) // 2
    p_want = torch.zeros(B * cols, heads, 2 * nlp, dtype=torch.float64)
    p_want[:, :, :Lq] = p_ref[..., 0]
    p_want[:, :, 1:Lq] -= p_ref[:, :, :1, 0]   # (levels 1.. carry their difference to level 0)
    p_want = torch.cat([p_want, torch.zeros_like(p_want)], dim=2)
    p_want[:, :, 2 * nlp:2 * nlp + Lq] = p_ref[..., 1]
    p_want[:, :, 2 * nlp + 1:2 * nlp + Lq] -= p_ref[:, :, :1, 1]
    assert (p_got[:, :, :4 * nlp] - p_want).abs().max().item() < 1e-6
    assert p_got[:, :, 4 * nlp:].abs().max().item() == 0
    # the values as fp16 pairs of (v0 - v2, v1 - v2, v2): exactly what the splitting kernel makes of them, rows (col, j)
    v = kvr[:, :, :, 1].permute(0, 2, 1, 3, 4).reshape(B * cols, Lk, inner).float().to(DEV)
    v = torch.stack([v[:, 0] - v[:, 2], v[:, 1] - v[:, 2], v[:, 2]], dim=1).reshape(B * cols * Lk, inner)
    assert torch.equal(Vp, L.split_f16(v))
    assert relerr(out, ref) < 3e-6


def _score_rows(q, w_kv, ctx, Lq, heads, hd):
    """fp64: the context rows as [v | scaled scores with every query | zero padding to a multiple of 4] -- what `to_kv` leaves
    when its key half is replaced by the rows W_k^T q_l / sqrt(head_dim) (model.hip:score_weights)."""
    inner = heads * hd
    w_s = torch.einsum("lhd,hdc->lhc", q.reshape(Lq, heads, hd), w_kv[:inner].reshape(heads, hd, -1)) / hd ** 0.5
    w_vs = torch.cat([w_kv[inner:], w_s.reshape(Lq * heads, -1)], dim=0)
    return ctx @ w_vs.T   # (rows, inner + Lq * heads)


# (the encoder's level aggregation: 3 latent queries over 13 levels; the decoder's de-aggregation: 13 level queries over 3
#  latents; a generic shape; head_dim 32)
@pytest.mark.parametrize("Lq,Lk,heads,hd,cols,B", [(3, 13, 16, 64, 70, 1), (13, 3, 16, 64, 41, 2), (5, 7, 4, 64, 33, 1), (4, 6, 8, 32, 29, 2)])
@pytest.mark.parametrize("pairs", [False, True])
def test_perceiver_attention_from_scores_equals_attention_on_keys(Lq, Lk, heads, hd, cols, B, pairs):
    """perceiver.py:141-152 with q . (W_k x) re-associated as (W_k^T q) . x: same softmax weights, same values."""
    L = lib()
    inner, D = heads * hd, 96
    q = rnd(Lq, inner, seed=1)
    w_kv = rnd(2 * inner, D, seed=2, scale=D ** -0.5)
    ctx = rnd(B * Lk * cols, D, seed=3)   # rows (b, j, col)
    kv = ctx @ w_kv.T
    vs = _score_rows(q, w_kv, ctx, Lq, heads, hd)
    ld = (vs.shape[1] + 7) // 8 * 8
    vs_dev = torch.zeros(vs.shape[0], ld, device=DEV)
    vs_dev[:, :vs.shape[1]] = vs.float().to(DEV)
    word = torch.tensor([1.0], device=DEV)
    guard = (word, 2.0) if pairs else None
    want = torch.full((B * cols * Lq, inner), float("nan"), device=DEV)
    L.perceiver_attention(q.float().to(DEV), 0, kv.float().to(DEV).contiguous(), want, B, cols, Lk * cols, cols, Lq, Lk, heads, hd,
                          pair_guard=guard)
    got = torch.full((B * cols * Lq, inner), float("nan"), device=DEV)
    L.perceiver_attention_scores(vs_dev, inner, got, B, cols, Lk * cols, cols, Lq, Lk, heads, hd, pair_guard=guard)
    torch.cuda.synchronize()
    if pairs:   # both wrote fp16 pairs: compare the values they stand for
        want, got = sum(_unsplit(want)), sum(_unsplit(got))
    kvr = kv.reshape(B, Lk, cols, 2, heads, hd).permute(3, 0, 2, 4, 1, 5)
    qq = q.reshape(Lq, heads, hd).permute(1, 0, 2)[None, None].expand(B, cols, -1, -1, -1)
    ref = F.scaled_dot_product_attention(qq, kvr[0], kvr[1]).permute(0, 1, 3, 2, 4).reshape(B * cols * Lq, inner)
    assert relerr(got, ref) < 2e-6 and relerr(want, ref) < 2e-6
    # a skip guard that holds retires the launch
    L.perceiver_attention_scores(vs_dev, inner, got.fill_(-7.0), B, cols, Lk * cols, cols, Lq, Lk, heads, hd, skip_guard=(word, 2.0))
    torch.cuda.synchronize()
    assert bool((got == -7.0).all())


@pytest.mark.parametrize("Lq,heads,cols,B", [(13, 16, 53, 1), (4, 2, 37, 2), (3, 4, 20, 1)])
def test_perceiver_probs_from_scores_equal_probs_on_keys(Lq, heads, cols, B):
    L = lib()
    Lk, hd, D = 3, 64, 128
    inner = heads * hd
    q = rnd(Lq, inner, seed=1)
    w_kv = rnd(2 * inner, D, seed=2, scale=D ** -0.5)
    ctx = rnd(B * Lk * cols, D, seed=3)
    kv = (ctx @ w_kv.T).float().to(DEV).contiguous()
    vs = _score_rows(q, w_kv, ctx, Lq, heads, hd)
    ld = (vs.shape[1] + 255) // 256 * 256   # (zero columns behind the scores, as the padded weight rows leave them)
    vs_dev = torch.zeros(vs.shape[0], ld, device=DEV)
    vs_dev[:, :vs.shape[1]] = vs.float().to(DEV)
    P0, V0 = L.perceiver_probs(q.float().to(DEV), kv, B, cols, Lk * cols, cols, Lq, Lk, heads, hd)
    P1, V1 = L.perceiver_probs_scores(vs_dev, inner, B, cols, Lk * cols, cols, Lq, Lk, heads, hd)
    torch.cuda.synchronize()
    assert torch.equal(V0, V1)   # the same value rows, split the same way
    assert (P0 - P1).abs().max().item() < 2e-6 and bool((P1[:, :, 32:] == 0).all())
    nlp = (Lq + 1) // 2
    assert bool((P1[:, :, 4 * nlp:32] == 0).all())


@pytest.mark.parametrize("holds", [True, False])
def test_perceiver_out_pair_and_plain_pair_follow_the_guard(holds):
    """The device word picks exactly one of {probs + out, attention + (three-term) to_out}; the other pair retires."""
    L = lib()
    B, cols, Lq, Lk, heads, hd, N = 1, 64, 13, 3, 16, 64, 1024
    inner = heads * hd
    q = rnd(Lq, inner, seed=1).float().to(DEV)
    kv = rnd(Lk * cols, 2 * inner, seed=2).float().to(DEV)
    word = torch.tensor([1.0 if holds else 3.0], device=DEV)
    P, Vp = L.perceiver_probs(q, kv, B, cols, Lk * cols, cols, Lq, Lk, heads, hd, guard=(word, 2.0))
    out = torch.full((cols * Lq, N), -7.0, device=DEV)
    w_pairs = L.split_f16(rnd(N, inner, seed=3, scale=inner ** -0.5).float().to(DEV), scale=64.0)
    L.perceiver_out(Vp, w_pairs, P, out, cols, Lq, Lk, heads, hd, guard=(word, 2.0))
    att = torch.full((cols * Lq, inner), -7.0, device=DEV)
    L.perceiver_attention(q, 0, kv, att, B, cols, cols * Lk, cols, Lq, Lk, heads, hd, skip_guard=(word, 2.0))
    torch.cuda.synchronize()
    assert bool((P != 0).any()) == holds and bool((out != -7.0).all()) == holds
    assert bool((att == -7.0).all()) == holds


def test_assemble_tokens():
    L = lib()
    B, Cl, Lp, D = 2, 4, 30, 64
    surf, agg = rnd(B * Lp, D, seed=1), rnd(B * Lp * (Cl - 1), D, seed=2)
    ps, te = rnd(Lp, D, seed=3), rnd(B, D, seed=4)
    out_f = torch.zeros((B * Cl * Lp, D), device=DEV)
    out_b = torch.zeros((B * Cl * Lp, D), dtype=torch.bfloat16, device=DEV)
    L.assemble_tokens(surf.float().to(DEV), agg.float().to(DEV), ps.float().to(DEV), te.float().to(DEV),
                      out_f, out_b, B, Cl, Lp, D)
    torch.cuda.synchronize()
    x = torch.cat([surf.reshape(B, 1, Lp, D), agg.reshape(B, Lp, Cl - 1, D).permute(0, 2, 1, 3)], dim=1)
    ref = (x + ps[None, None] + te[:, None, None]).reshape(-1, D)
    assert relerr(out_f, ref) < 1e-6
    assert relerr(out_b.float(), ref) < 5e-3


def test_convert_and_copy2d():
    L = lib()
    x = rnd(1000, 37, seed=1).float().to(DEV).contiguous()
    y = L.convert(x, torch.empty_like(x, dtype=torch.bfloat16))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.cpu().bfloat16())  # round-to-nearest-even, bit exact
    z = L.convert(y, torch.empty_like(x))
    torch.cuda.synchronize()
    assert torch.equal(z.cpu(), y.cpu().float())
    src = rnd(20, 64, seed=2).float().to(DEV)
    dst = torch.zeros((20, 128), device=DEV)
    L.copy2d(src, dst[:, 64:])
    torch.cuda.synchronize()
    assert torch.equal(dst[:, 64:], src) and (dst[:, :64] == 0).all()


@pytest.mark.parametrize("n", [4, 1000, 4097, 3_000_001])
def test_absmax_ignores_nan_and_finds_the_largest_magnitude(n):
    from aurora_amd.engine import lib as L

    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, device="cuda", generator=g)
    x[n // 3] = -123.5
    if n > 8:
        x[n - 2] = float("nan")
    out = L.absmax(x)
    torch.cuda.synchronize()
    assert out.item() == 123.5
    x[0] = float("-inf")
    assert L.absmax(x).item() == float("inf")
