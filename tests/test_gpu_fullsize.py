"""BASELINE-size checks (0.25 degree: 4 x 180 x 360 tokens) through size-independent properties --
the CPU oracle would need minutes per evaluation at these sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from aurora_amd.engine import lib

    lib.load()
    return lib


@pytest.mark.parametrize("stage", [0, 2])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_full_grid_properties(stage, shifted):
    """At full size: (1) linear in V, (2) a window's output is a convex combination of that
    window's V rows (bounded by their min/max), (3) every real token is written exactly once."""
    from aurora_amd.engine import geometry

    L_ = _lib()
    res, D, heads = [((4, 180, 360), 512, 8), None, ((4, 45, 90), 2048, 32)][stage]
    L = res[0] * res[1] * res[2]
    tok, grp, _ = geometry.window_tables(res, (2, 6, 12), shifted)
    tok_d = torch.from_numpy(tok).to(DEV)
    grp_d = None if grp is None else torch.from_numpy(grp).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(stage * 2 + shifted)
    qk = torch.randn(L, 2 * D, device=DEV, generator=g)
    v1, v2 = torch.randn(L, D, device=DEV, generator=g), torch.randn(L, D, device=DEV, generator=g)
    bias = torch.zeros(3 * D, device=DEV)

    def run(v):
        qkv = torch.cat([qk, v], dim=1).bfloat16().contiguous()
        out = torch.full((L, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        L_.window_attention(qkv, bias, out, tok_d, grp_d, 1, L, D, heads)
        return out.float()

    o1, o2, o12 = run(v1), run(v2), run(v1 + v2)
    torch.cuda.synchronize()
    assert torch.isfinite(o12).all()  # every token written (NaN pre-fill gone)
    err = (o12 - (o1 + o2)).abs().max() / o12.abs().max()
    assert err < 3e-2, err  # bf16 inputs/outputs: three roundings
    # convexity per (window, head): outputs within [min, max] of the window's V values (+ bf16 slack)
    w = 7
    rows = tok_d[w][tok_d[w] >= 0].long()
    vwin = torch.cat([v1[rows].bfloat16().float(), torch.zeros(1, D, device=DEV)])  # padded rows carry v = bias = 0
    lo, hi = vwin.reshape(-1, heads, 64).amin(0), vwin.reshape(-1, heads, 64).amax(0)
    ow = o1[rows].reshape(-1, heads, 64)
    assert (ow >= lo - 0.05).all() and (ow <= hi + 0.05).all()


def test_linear_full_size_matches_chunked_rows():
    """259,200 x 2048 x 512 bf16 GEMM (stage-0 fc1): the big-tile kernel on the whole matrix equals the
    same kernel family applied to row chunks (different tile assignments, same arithmetic order in K)."""
    L_ = _lib()
    M, N, K = 259200, 2048, 512
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=DEV, generator=g)
    full = L_.linear(a, w, b, torch.empty(M, N, dtype=torch.bfloat16, device=DEV), act=L_.ACT_GELU)
    ref = torch.empty_like(full)
    for s in range(0, M, 777):  # ragged chunks: the small kernel below 1024 rows
        e = min(M, s + 777)
        L_.linear(a[s:e], w, b, ref[s:e], act=L_.ACT_GELU)
    torch.cuda.synchronize()
    assert torch.equal(full, ref)


def test_full_size_step_bf16_vs_fp32_engine_and_determinism():
    """One 721 x 1440 x 13 step of a depth-reduced 0.25-degree model: the bf16 engine stays within the
    autocast tolerance of the fp32 engine, and two runs are bit-identical."""
    import aurora_amd
    from bench import synthetic_batch

    kw = dict(encoder_depths=(2, 2, 2), decoder_depths=(2, 2, 2), use_lora=False)
    outs = {}
    for autocast in (False, True):
        torch.manual_seed(0)
        with torch.device(DEV):
            model = aurora_amd.Aurora(autocast=autocast, **kw)
            with torch.no_grad():
                for p in model.parameters():
                    if not p.any():
                        p.normal_(std=0.02)
        model.eval()
        batch = synthetic_batch(model.config, 721, 1440, 1, DEV)
        with torch.inference_mode():
            p1 = model.forward(batch)
            p2 = model.forward(batch)
        torch.cuda.synchronize()
        assert p1.surf_vars["2t"].shape == (1, 1, 720, 1440) and p1.atmos_vars["q"].shape == (1, 1, 13, 720, 1440)
        for k in p1.atmos_vars:
            assert torch.equal(p1.atmos_vars[k], p2.atmos_vars[k])
        outs[autocast] = {**{f"s.{k}": v.cpu() for k, v in p1.surf_vars.items()},
                          **{f"a.{k}": v.cpu() for k, v in p1.atmos_vars.items()}}
        del model
        torch.cuda.empty_cache()
    for k, ref in outs[False].items():
        err = (outs[True][k] - ref).abs().mean() / ref.abs().mean()
        assert err < 3e-2, (k, err.item())
