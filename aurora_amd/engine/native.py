"""The C-ABI model handle (`aurora_hip_create` ... `aurora_hip_step`, include/aurora_hip.h) driven from Python.

`NativeModel` is all the Python there is around a forecast step of the ERA5 model family: it hands the configuration
and the state_dict to the handle once, the grid-dependent tables once per grid, and per step only raw device pointers.
The launch sequence of the step lives in aurora_amd/csrc/model.hip; any other host language binds the same seven
functions.  `Engine.step` routes here when `supports(model)`; the air-pollution / ocean-wave variants
and latitude-band sharded steps are still sequenced by engine.py over the operator entry points.
"""
from __future__ import annotations

import ctypes
from datetime import timedelta

import numpy as np
import torch

from aurora_amd import normalisation
from aurora_amd.batch import Batch, derive_metadata
from aurora_amd.engine import encodings, lib

_LORA_MODES = {"single": 0, "from_second": 1, "all": 2}


def supports(model) -> bool:
    """Is this model inside the scope of the C-ABI step (csrc/model.hip)?"""
    cfg = model.config
    return (model.variant == "base" and not cfg.level_condition and not cfg.dynamic_vars and not cfg.atmos_static_vars
            and not cfg.dec_separate_perceiver and not cfg.modulation_heads and not cfg.positive_surf_vars
            and not cfg.positive_atmos_vars and not cfg.simulate_indexing_bug and len(cfg.encoder_depths) <= 4
            and len(cfg.encoder_depths) == len(cfg.decoder_depths) and cfg.lora_mode in _LORA_MODES)


def _version(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:   # inference tensors carry no version counter
        return -1


def _strs(names):
    arr = (ctypes.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    return arr


def _dbl(values):
    return (ctypes.c_double * max(len(values), 1))(*[float(v) for v in values])


class NativeModel:
    def __init__(self, model) -> None:
        L = lib.load()
        cfg = model.config
        assert supports(model)
        self.cfg, self.model = cfg, model
        n = len(cfg.encoder_depths)
        c = lib.HipConfig()
        c.embed_dim, c.patch_size, c.latent_levels, c.num_heads = cfg.embed_dim, cfg.patch_size, cfg.latent_levels, cfg.num_heads
        c.n_stages = n
        for i in range(n):
            c.encoder_depths[i], c.encoder_heads[i] = cfg.encoder_depths[i], cfg.encoder_num_heads[i]
            c.decoder_depths[i], c.decoder_heads[i] = cfg.decoder_depths[i], cfg.decoder_num_heads[i]
        for i in range(3):
            c.window[i] = cfg.window_size[i]
        c.enc_depth, c.dec_depth, c.perceiver_ln_eps = cfg.enc_depth, cfg.dec_depth, cfg.perceiver_ln_eps
        c.max_history = cfg.max_history_size
        c.timestep_hours = cfg.timestep / timedelta(hours=1)
        c.stabilise_level_agg, c.use_lora = int(cfg.stabilise_level_agg), int(cfg.use_lora)
        c.lora_steps, c.lora_mode, c.autocast = cfg.lora_steps, _LORA_MODES[cfg.lora_mode], int(model.autocast)
        self._names = (_strs(cfg.surf_vars), _strs(cfg.static_vars), _strs(cfg.atmos_vars))   # kept alive
        c.n_surf, c.n_static, c.n_atmos = len(cfg.surf_vars), len(cfg.static_vars), len(cfg.atmos_vars)
        c.surf_vars, c.static_vars, c.atmos_vars = self._names
        handle = ctypes.c_void_p()
        lib._check(L.aurora_hip_create(ctypes.byref(c), ctypes.byref(handle)))
        self._h = handle
        for name, t in model.state_dict().items():
            t = t.detach().contiguous()
            assert t.dtype == torch.float32
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            lib._check(L.aurora_hip_pack_weights(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim(),
                                                 lib.F32, int(t.is_cuda)))
        lib._check(L.aurora_hip_finalize(self._h, lib._stream()))
        self._grid_key = None
        self._grid_ident = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib.load().aurora_hip_destroy(h)
            except Exception:   # interpreter shutdown
                pass

    def save_packed(self, path: str) -> None:
        """Write the handle's weights as a packed file (include/aurora_hip.h: aurora_hip_save_packed): what a host
        without Python loads with `aurora_hip_load_packed` instead of unpickling a checkpoint."""
        lib._check(lib.load().aurora_hip_save_packed(self._h, str(path).encode(), lib._stream()))

    # -- per grid / level set ---------------------------------------------------------------------------------
    def _precompute(self, lat: torch.Tensor, lon: torch.Tensor, levels: tuple, H: int, W: int) -> None:
        # storage identity first (no device->host copy per step), then content
        ident = (lat.data_ptr(), tuple(lat.shape), _version(lat), lon.data_ptr(), tuple(lon.shape), _version(lon), levels,
                 tuple(sorted(self.model.surf_stats.items())))
        if ident == self._grid_ident:
            return
        lat_h, lon_h = lat.detach().cpu(), lon.detach().cpu()
        key = (lat_h.numpy().tobytes(), lon_h.numpy().tobytes(), tuple(lat_h.shape), tuple(lon_h.shape), ident[6], ident[7])
        if key != self._grid_key:
            cfg = self.cfg
            # The fp32 geometry of the reference's position / scale encodings runs through torch's own CPU kernels
            # (encodings.py explains why); the handle takes the resulting tables.
            pos, scale = encodings.pos_scale_encodings(cfg.embed_dim, lat_h, lon_h, cfg.patch_size)
            pos, scale = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(scale, np.float32)
            g = lib.HipGrid()
            g.n_lat, g.n_lon = H, W
            g.n_levels = len(levels)
            lv = _dbl(levels)
            g.levels = lv
            g.levels_float32 = int(not all(isinstance(v, (int, np.integer)) for v in levels))
            s_aff = [normalisation.surf_affine(n, self.model.surf_stats) for n in cfg.surf_vars]
            t_aff = [normalisation.surf_affine(n, self.model.surf_stats) for n in cfg.static_vars]
            a_aff = [normalisation.atmos_affine(n, levels) for n in cfg.atmos_vars]
            keep = [lv, _dbl([a[0] for a in s_aff]), _dbl([a[1] for a in s_aff]), _dbl([a[0] for a in t_aff]),
                    _dbl([a[1] for a in t_aff]), _dbl([x for a in a_aff for x in a[0]]),
                    _dbl([x for a in a_aff for x in a[1]])]
            g.surf_loc, g.surf_scale, g.static_loc, g.static_scale, g.atmos_loc, g.atmos_scale = keep[1:]
            g.pos_encoding = pos.ctypes.data_as(lib._PF)
            g.scale_encoding = scale.ctypes.data_as(lib._PF)
            lib._check(lib.load().aurora_hip_precompute(self._h, ctypes.byref(g), lib._stream()))
            self._grid_key = key
        self._grid_ident = ident
        self._keep_coords = (lat, lon)

    # -- per-launch timing (HIP events inside the handle, on the launch stream) -------------------------------------
    def profile_begin(self, only=None) -> None:
        mask = 0
        for i, k in enumerate(lib.PROFILE_KINDS):
            if only is None or k in only:
                mask |= 1 << i
        lib._check(lib.load().aurora_hip_profile_begin(self._h, mask))

    def profile_end(self) -> dict:
        """{kernel: {"launches", "ms", "work"}} of the launches since `profile_begin` (synchronises the device)."""
        out = (lib.HipProfileEntry * len(lib.PROFILE_KINDS))()
        n = ctypes.c_int(0)
        lib._check(lib.load().aurora_hip_profile_end(self._h, out, len(out), ctypes.byref(n)))
        return {e.kernel.decode(): {"launches": e.launches, "ms": e.ms, "work": e.work} for e in out[:n.value] if e.launches}

    def set_time(self, times) -> None:
        stamps = _dbl([t.timestamp() / 3600 for t in times])
        lib._check(lib.load().aurora_hip_set_time(self._h, stamps, len(times), lib._stream()))

    # -- the step ---------------------------------------------------------------------------------------------
    def accepts(self, batch: Batch) -> bool:
        cfg = self.cfg
        return (set(batch.surf_vars) == set(cfg.surf_vars) and set(batch.static_vars) == set(cfg.static_vars)
                and set(batch.atmos_vars) == set(cfg.atmos_vars) and batch.metadata.lat.dim() == 1)

    @torch.no_grad()
    def step(self, batch: Batch, upload_time: bool = True, out=None) -> Batch:
        """`batch`: float32, cropped to the patch size, on the device.  `out`: see `Aurora.forward`."""
        cfg = self.cfg
        md = batch.metadata
        levels = tuple(md.atmos_levels)
        H, W = batch.spatial_shape
        B, T = next(iter(batch.surf_vars.values())).shape[:2]
        assert md.lat.shape[0] == H and md.lon.shape[-1] == W
        assert md.lat.dtype in (torch.float32, torch.float64), f"Latitude num. unstable: {md.lat.dtype}."
        assert md.lon.dtype in (torch.float32, torch.float64), f"Longitude num. unstable: {md.lon.dtype}."
        self._precompute(md.lat, md.lon, levels, H, W)
        if upload_time:
            self.set_time(md.time)

        def same_strides(ts):
            ts = [t if t.dtype == torch.float32 else t.float() for t in ts]
            if len({t.stride() for t in ts}) > 1 or any(t.stride(-1) != 1 for t in ts):
                ts = [t.contiguous() for t in ts]
            return ts

        surf = same_strides([batch.surf_vars[n] for n in cfg.surf_vars])
        stat = same_strides([batch.static_vars[n] for n in cfg.static_vars])
        atmos = same_strides([batch.atmos_vars[n] for n in cfg.atmos_vars])
        dev = surf[0].device
        def dest(kind, name, shape):
            t = None if out is None else out[kind].get(name)
            if (t is not None and tuple(t.shape) == shape and t.dtype == torch.float32 and t.is_contiguous()
                    and t.device == dev):
                return t
            return torch.empty(shape, dtype=torch.float32, device=dev)

        out_s = [dest(0, n, (B, 1, H, W)) for n in cfg.surf_vars]
        out_a = [dest(1, n, (B, 1, len(levels), H, W)) for n in cfg.atmos_vars]
        ptrs = lambda ts: (ctypes.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])  # noqa: E731
        io = lib.HipStepIO()
        io.B, io.T, io.rollout_step = B, T, md.rollout_step
        keep = [ptrs(surf), ptrs(stat), ptrs(atmos), ptrs(out_s), ptrs(out_a)]
        io.surf, io.stat, io.atmos, io.out_surf, io.out_atmos = keep
        io.surf_strides[:] = surf[0].stride()
        if stat:
            io.static_strides[:] = stat[0].stride()
        io.atmos_strides[:] = atmos[0].stride()
        lib._check(lib.load().aurora_hip_step(self._h, ctypes.byref(io), lib._stream()))
        self._keepalive = (surf, stat, atmos)     # inputs must outlive the enqueued kernels
        order_s = {n: i for i, n in enumerate(cfg.surf_vars)}
        order_a = {n: i for i, n in enumerate(cfg.atmos_vars)}
        new_md = derive_metadata(md, lat=md.lat.to(torch.float32), lon=md.lon.to(torch.float32),
                                 time=tuple(t + cfg.timestep for t in md.time), rollout_step=md.rollout_step + 1)
        return Batch({n: out_s[order_s[n]] for n in batch.surf_vars}, dict(batch.static_vars),
                     {n: out_a[order_a[n]] for n in batch.atmos_vars}, new_md)
