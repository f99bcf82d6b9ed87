"""A forecast through the C ABI alone: create / pack_weights / finalize / precompute / set_time / step driven with
ctypes, numpy host arrays and raw device pointers -- no `Engine`, no `aurora_amd.Aurora.forward`, no Python
sequencing.  torch appears only as the device allocator (and for the history shift between roll-out steps, which is the
caller's job exactly as in aurora/rollout.py:39-49).  The predictions must match the reference goldens to the same
tolerance as the Python-facing API (tests/test_gpu_model.py): mean-rel <= 1e-4 per variable.

The handle computes the Fourier position / scale tables itself here (lat / lon passed, no tables): the C++ restatement of
posencoding.py:61-192 is what is being checked.  The air-pollution and ocean-wave variants go through the same seven
functions (variant keywords in aurora_hip_config).
"""
import ctypes

import numpy as np
import pytest
import torch

import aurora_amd
from aurora_amd import normalisation
from aurora_amd.engine import lib
from tests import helpers
from tests.golden_cases import CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"
_MODES = {"single": 0, "from_second": 1, "all": 2}


def _carr(ctype, values):
    return (ctype * max(len(values), 1))(*values)


class Handle:
    """The binding a maintainer of another host language would write, in ctypes."""

    def __init__(self, cfg, autocast, state_dict=None, packed=None, variant=None):
        self.L = lib.load()
        self.cfg = cfg
        c = lib.HipConfig()
        c.embed_dim, c.patch_size, c.latent_levels, c.num_heads = cfg.embed_dim, cfg.patch_size, cfg.latent_levels, cfg.num_heads
        c.n_stages = len(cfg.encoder_depths)
        for i in range(c.n_stages):
            c.encoder_depths[i], c.encoder_heads[i] = cfg.encoder_depths[i], cfg.encoder_num_heads[i]
            c.decoder_depths[i], c.decoder_heads[i] = cfg.decoder_depths[i], cfg.decoder_num_heads[i]
        for i in range(3):
            c.window[i] = cfg.window_size[i]
        c.enc_depth, c.dec_depth, c.perceiver_ln_eps = cfg.enc_depth, cfg.dec_depth, cfg.perceiver_ln_eps
        c.max_history, c.timestep_hours = cfg.max_history_size, cfg.timestep.total_seconds() / 3600
        c.stabilise_level_agg, c.use_lora, c.lora_steps = int(cfg.stabilise_level_agg), int(cfg.use_lora), cfg.lora_steps
        c.lora_mode, c.autocast = _MODES[cfg.lora_mode], int(autocast)
        strs = lambda v: _carr(ctypes.c_char_p, [n.encode() for n in v])  # noqa: E731
        self._names = [strs(v) for v in (cfg.surf_vars, cfg.static_vars, cfg.atmos_vars)]
        c.n_surf, c.n_static, c.n_atmos = len(cfg.surf_vars), len(cfg.static_vars), len(cfg.atmos_vars)
        c.surf_vars, c.static_vars, c.atmos_vars = self._names
        # variant keywords (aurora.py:86-95); `variant` = (code, extra) as a non-Python host would hard-wire them
        self.surf_inputs = tuple(cfg.surf_vars)
        if variant is not None:
            code, extra = variant
            c.variant = code
            lc = tuple(cfg.level_condition or ())
            diff = extra.get("difference_history", {})
            self._vkeep = [_carr(ctypes.c_double, [float(x) for x in lc]), strs(cfg.separate_perceiver), strs(cfg.modulation_heads),
                           _carr(ctypes.c_int32, [diff.get(v, -1) for v in cfg.modulation_heads]), strs(cfg.positive_surf_vars),
                           strs(cfg.positive_atmos_vars), strs(extra.get("surf_inputs", ())), strs(extra.get("density", ())),
                           strs(extra.get("angle", ()))]
            c.n_level_condition, c.level_condition = len(lc), self._vkeep[0]
            c.dynamic_vars, c.atmos_static_vars = int(cfg.dynamic_vars), int(cfg.atmos_static_vars)
            c.clamp_at_first_step, c.simulate_indexing_bug = int(cfg.clamp_at_first_step), int(cfg.simulate_indexing_bug)
            c.n_separate_perceiver, c.separate_perceiver = len(cfg.separate_perceiver), self._vkeep[1]
            c.n_modulation_heads, c.modulation_heads, c.difference_history = len(cfg.modulation_heads), self._vkeep[2], self._vkeep[3]
            c.n_positive_surf, c.positive_surf_vars = len(cfg.positive_surf_vars), self._vkeep[4]
            c.n_positive_atmos, c.positive_atmos_vars = len(cfg.positive_atmos_vars), self._vkeep[5]
            if extra.get("surf_inputs"):
                self.surf_inputs = tuple(extra["surf_inputs"])
                c.n_surf_inputs, c.surf_inputs = len(self.surf_inputs), self._vkeep[6]
                c.n_density, c.density_channel_surf_vars = len(extra["density"]), self._vkeep[7]
                c.n_angle, c.angle_surf_vars = len(extra["angle"]), self._vkeep[8]
        self.h = ctypes.c_void_p()
        self.check(self.L.aurora_hip_create(ctypes.byref(c), ctypes.byref(self.h)))
        if packed is not None:                       # a packed weight file instead of a state_dict
            self.check(self.L.aurora_hip_load_packed(self.h, str(packed).encode()))
        for name, w in (state_dict or {}).items():   # HOST arrays: on_device = 0
            w = np.ascontiguousarray(w, np.float32)
            shape = _carr(ctypes.c_int64, list(w.shape))
            self.check(self.L.aurora_hip_pack_weights(self.h, name.encode(), w.ctypes.data_as(ctypes.c_void_p), shape, w.ndim, 0, 0))
        self.check(self.L.aurora_hip_finalize(self.h, None))
        n_out = self.L.aurora_hip_output_vars(self.h, None, 0)
        names = (ctypes.c_char_p * n_out)()
        self.L.aurora_hip_output_vars(self.h, names, n_out)
        self.surf_outputs = tuple(x.decode() for x in names)

    def check(self, code):
        assert code == 0, self.L.aurora_hip_last_error().decode()

    def precompute(self, lat, lon, levels):
        cfg, g = self.cfg, lib.HipGrid()
        g.n_lat, g.n_lon, g.n_levels = len(lat), len(lon), len(levels)
        dbl = lambda v: _carr(ctypes.c_double, [float(x) for x in v])  # noqa: E731
        sa = [normalisation.surf_affine(n) for n in self.surf_inputs]
        ta = [normalisation.surf_affine(n) for n in cfg.static_vars]
        aa = [normalisation.atmos_affine(n, levels) for n in cfg.atmos_vars]
        keep = [dbl(lat), dbl(lon), dbl(levels), dbl([a[0] for a in sa]), dbl([a[1] for a in sa]), dbl([a[0] for a in ta]),
                dbl([a[1] for a in ta]), dbl([x for a in aa for x in a[0]]), dbl([x for a in aa for x in a[1]])]
        (g.lat, g.lon, g.levels, g.surf_loc, g.surf_scale, g.static_loc, g.static_scale, g.atmos_loc, g.atmos_scale) = keep
        g.levels_float32 = int(not all(isinstance(v, int) for v in levels))
        self.check(self.L.aurora_hip_precompute(self.h, ctypes.byref(g), None))

    def step(self, surf, static, atmos, times, rollout_step, levels):
        """surf / atmos: lists of device tensors (B, T, H, W) / (B, T, C, H, W); returns lists of (B, H', W) / (B, C, H', W)."""
        cfg = self.cfg
        B, T, H, W = surf[0].shape
        Hc = H - H % cfg.patch_size
        hours = _carr(ctypes.c_double, [t.timestamp() / 3600 for t in times])
        cal = _carr(ctypes.c_int32, [x for t in times for x in (t.hour, t.weekday(), t.day)])
        self.check(self.L.aurora_hip_set_time_ex(self.h, hours, cal, B, None))
        out_s = [torch.empty(B, Hc, W, device=DEV) for _ in self.surf_outputs]
        out_a = [torch.empty(B, len(levels), Hc, W, device=DEV) for _ in atmos]
        io = lib.HipStepIO()
        io.B, io.T, io.rollout_step = B, T, rollout_step
        ptr = lambda ts: _carr(ctypes.c_void_p, [t.data_ptr() for t in ts])  # noqa: E731
        keep = [ptr(surf), ptr(static), ptr(atmos), ptr(out_s), ptr(out_a)]
        io.surf, io.stat, io.atmos, io.out_surf, io.out_atmos = keep
        io.surf_strides[:] = surf[0].stride()
        io.static_strides[:] = static[0].stride()
        io.atmos_strides[:] = atmos[0].stride()
        self.check(self.L.aurora_hip_step(self.h, ctypes.byref(io), None))   # stream 0
        torch.cuda.synchronize()
        return out_s, out_a

    def close(self):
        self.L.aurora_hip_destroy(self.h)


@pytest.mark.parametrize("name", ["base_pad", "small_b2", "lora_all", "stabilised_12h", "patch10"])
def test_c_abi_rollout_matches_reference_golden(name):
    case, meta = helpers.case_model_meta(name)
    cfg = meta.config
    sd = {k: v.numpy() for k, v in helpers.case_state_dict(meta, torch.float32).items()}
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    gold = helpers.load_golden(name)
    levels = tuple(case["levels"])
    h = Handle(cfg, autocast=False, state_dict=sd)
    h.precompute(lat.tolist(), lon.tolist(), levels)
    f = lambda d, names: [d[n].float().to(DEV).contiguous() for n in names]  # noqa: E731
    s_in, t_in, a_in = f(surf, cfg.surf_vars), f(static, cfg.static_vars), f(atmos, cfg.atmos_vars)
    P = cfg.patch_size
    Hc = case["H"] - case["H"] % P
    worst = 0.0
    for step in range(case["steps"]):
        out_s, out_a = h.step(s_in, t_in, a_in, times, step, levels)
        for kind, names, outs in (("surf", cfg.surf_vars, out_s), ("atmos", cfg.atmos_vars, out_a)):
            for n, o in zip(names, outs):
                ref = torch.from_numpy(gold[f"s{step}.{kind}.{n}"])      # (B, 1, [C,] H', W)
                got = o.cpu().reshape(ref.shape)
                e = helpers.mean_rel_err(got, ref)
                worst = max(worst, e)
                assert e <= 1e-4 and helpers.rel_err(got, ref) <= 1e-3, (step, kind, n, e)
        # the caller's side of rollout(): drop the oldest state, append the prediction (rollout.py:39-49)
        s_in = [torch.cat([x[:, 1:, :Hc], o[:, None]], dim=1).contiguous() for x, o in zip(s_in, out_s)]
        a_in = [torch.cat([x[:, 1:, :, :Hc], o[:, None]], dim=1).contiguous() for x, o in zip(a_in, out_a)]
        t_in = [x[:Hc].contiguous() for x in t_in]
        times = tuple(t + cfg.timestep for t in times)
        if step == 0 and case["H"] != Hc:
            h.precompute(lat[:Hc].tolist(), lon.tolist(), levels)     # the cropped grid from now on
    print(name, "C-ABI roll-out worst mean-rel", worst)
    h.close()


@pytest.mark.parametrize("name", ["base_pad", "lora_all"])
def test_packed_weight_file_round_trip(name, tmp_path):
    """aurora_hip_save_packed -> aurora_hip_load_packed into a fresh handle: bit-identical bf16-backbone predictions,
    with the big backbone matrices stored in bf16 (no LoRA: attention projections too) and no state_dict in sight."""
    case, meta = helpers.case_model_meta(name)
    cfg = meta.config
    sd = {k: v.numpy() for k, v in helpers.case_state_dict(meta, torch.float32).items()}
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    levels = tuple(case["levels"])
    f = lambda d, names: [d[n].float().to(DEV).contiguous() for n in names]  # noqa: E731
    s_in, t_in, a_in = f(surf, cfg.surf_vars), f(static, cfg.static_vars), f(atmos, cfg.atmos_vars)
    a = Handle(cfg, autocast=True, state_dict=sd)
    path = tmp_path / "weights.aurorahip"
    a.check(a.L.aurora_hip_save_packed(a.h, str(path).encode(), None))
    n_params = sum(int(np.prod(v.shape)) for v in sd.values())
    assert path.stat().st_size < 4 * n_params * 0.8          # most of a tiny model's bytes are backbone matrices
    b = Handle(cfg, autocast=True, packed=path)
    outs = []
    for h in (a, b):
        h.precompute(lat.tolist(), lon.tolist(), levels)
        outs.append(h.step(s_in, t_in, a_in, times, 0, levels))
    for x, y in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
        assert torch.equal(x, y)
    # a bf16-only file cannot serve an fp32 backbone: said clearly
    try:
        Handle(cfg, autocast=False, packed=path)
        raised = False
    except AssertionError as e:
        raised = "bf16 only" in str(e)
    assert raised
    a.close()
    b.close()


def _variant_of(name, meta):
    if name == "air_pollution":
        return 1, {"difference_history": dict(type(meta)._predict_difference_history_dim_lookup)}
    if name == "wave":
        from aurora_amd.engine import native

        return 2, {"surf_inputs": native.wave_sources(meta), "density": meta.density_channel_surf_vars,
                   "angle": meta.angle_surf_vars}
    return None


@pytest.mark.parametrize("name", ["air_pollution", "wave"])
def test_c_abi_variants_match_reference_golden(name):
    """AuroraAirPollution (level-conditioned embeddings / heads, dynamic + static inputs, feature combiners, difference
    prediction, second decoder Perceiver) and AuroraWave (density / sin / cos channels and their inverse) through the raw C
    ABI: first roll-out step of the reference goldens.  (The wave variant's raw-batch preparation -- wind components, NaN
    marking of absent systems, aurora.py:854-890 -- is host-side input preparation: `aurora_amd.model.wave`.)"""
    from aurora_amd import Batch, Metadata
    from aurora_amd.model import wave

    case, meta = helpers.case_model_meta(name)
    cfg = meta.config
    sd = {k: v.numpy() for k, v in helpers.case_state_dict(meta, torch.float32).items()}
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    if name == "wave":
        surf = wave.transform_batch(Batch(surf, static, atmos, Metadata(lat, lon, times, tuple(case["levels"])))).surf_vars
    gold = helpers.load_golden(name)
    levels = tuple(case["levels"])
    h = Handle(cfg, autocast=False, state_dict=sd, variant=_variant_of(name, meta))
    h.precompute(lat.tolist(), lon.tolist(), levels)
    f = lambda d, names: [d[n].float().to(DEV).contiguous() for n in names]  # noqa: E731
    out_s, out_a = h.step(f(surf, h.surf_inputs), f(static, cfg.static_vars), f(atmos, cfg.atmos_vars), times, 0, levels)
    worst = 0.0
    for kind, names, outs in (("surf", h.surf_outputs, out_s), ("atmos", cfg.atmos_vars, out_a)):
        for n, o in zip(names, outs):
            ref = torch.from_numpy(gold[f"s0.{kind}.{n}"])
            got, ref, flipped = helpers.nan_agreement(o.cpu().reshape(ref.shape), ref)
            assert flipped <= 2e-3, (n, flipped)
            e = helpers.mean_rel_err(got, ref)
            worst = max(worst, e)
            assert e <= 1e-4 and helpers.rel_err(got, ref) <= 1e-3, (kind, n, e)
    assert len(h.surf_outputs) + len(cfg.atmos_vars) == sum(k.startswith("s0.") for k in gold)
    print(name, "C-ABI step worst mean-rel", worst)
    h.close()


def test_c_abi_argument_errors():
    L = lib.load()
    c = lib.HipConfig()
    h = ctypes.c_void_p()
    assert L.aurora_hip_create(ctypes.byref(c), ctypes.byref(h)) == -1      # zero stages
    assert b"stages" in L.aurora_hip_last_error()
    assert L.aurora_hip_step(None, None, None) == -1


@pytest.mark.parametrize("mode", ["bf16", "native", "f16"])
def test_c_abi_rollout_under_a_pinned_fp32_gemm_mode(mode):
    """AURORA_F32_GEMM pins how the large fp32 linears are multiplied (include/aurora_hip.h) -- a process-wide default, so
    the goldens are re-run in a child process per mode.  (A pinned mode once made the handle's resampler pass a guard
    to mode-1 launches, which then skipped to_kv / to_out: the results were garbage, and no test set the variable.)"""
    import os
    import subprocess
    import sys

    res = subprocess.run(
        [sys.executable, "-m", "pytest", __file__, "-q", "-x", "-p", "no:cacheprovider", "-k",
         "test_c_abi_rollout_matches_reference_golden and (base_pad or stabilised_12h)"],
        env={**os.environ, "AURORA_F32_GEMM": mode}, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "2 passed" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
