"""The C-ABI model handle (`aurora_hip_create` ... `aurora_hip_step`, include/aurora_hip.h) driven from Python.

`NativeModel` is all the Python there is around a forecast step: it hands the configuration and the state_dict to the
handle once, the grid-dependent tables once per grid, and per step only raw device pointers.  The launch sequence of the
step -- every model class, one device or one latitude band of a sharded forecast -- lives in aurora_amd/csrc/step.hip;
any other host language binds the same functions.

Sharded forecasts: the handle gathers the halo rows of a shifted-window block into staging buffers and calls back into
`_Transport` here, which moves them with torch.distributed point-to-point operations (backend "nccl" = RCCL over xGMI;
"gloo" host-staged for tests that run several ranks on one GPU).
"""
from __future__ import annotations

import ctypes
from datetime import timedelta

import numpy as np
import torch

from aurora_amd import normalisation
from aurora_amd.batch import BandBatch, Batch, derive_metadata
from aurora_amd.engine import encodings, lib

_LORA_MODES = {"single": 0, "from_second": 1, "all": 2}
_VARIANTS = {"base": 0, "air_pollution": 1, "wave": 2}


def _version(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:   # inference tensors carry no version counter
        return -1


def _strs(names):
    return (ctypes.c_char_p * max(len(names), 1))(*[n.encode() for n in names])


def _dbl(values):
    return (ctypes.c_double * max(len(values), 1))(*[float(v) for v in values])


def _i32(values):
    return (ctypes.c_int32 * max(len(values), 1))(*[int(v) for v in values])


def wave_sources(model) -> tuple[str, ...]:
    """The surface variables an ocean-wave model is fed (after `batch_transform_hook`), in the order of its channels:
    the model's `surf_vars` are channel names (`<v>_sin`, `<v>_cos`, `<v>_density`, aurora.py:854-890 upstream)."""
    out: list[str] = []
    for name in model.config.surf_vars:
        base = name
        for suffix, owners in (("_sin", model.angle_surf_vars), ("_cos", model.angle_surf_vars),
                               ("_density", model.density_channel_surf_vars)):
            if name.endswith(suffix) and name[:-len(suffix)] in owners:
                base = name[:-len(suffix)]
        if base not in out:
            out.append(base)
    return tuple(out)


class _Transport:
    """Halo transport of a sharded step: the `post` / `wait` callbacks the handle calls (include/aurora_hip.h).

    The two staging buffers (what is sent, what is received) are torch tensors, so that torch.distributed can address
    them; a message is a byte range of one of them.  `post` starts the point-to-point operations of one exchange -- grouped into one ncclGroup on RCCL's
    own stream, so the interior windows launched meanwhile overlap the transfer -- and `wait` makes the launch stream
    wait for them (not the host)."""

    def __init__(self, shard, device) -> None:
        self.shard, self.device = shard, device
        self.send = self.recv = None
        self.works: list = []
        self.error: BaseException | None = None
        self.post_c = lib.HALO_POST_FN(self._post)
        self.wait_c = lib.HALO_WAIT_FN(self._wait)
        # optional timing of `wait` (bench.py --gpus N): an event pair on the launch stream around every wait, i.e. how long
        # that stream stood still for messages that had not arrived when the halo projection was due
        self.time_waits = False
        self._wait_events: list = []
        self.exchanges = 0

    def allocate(self, n_bytes: int) -> None:
        n = max(int(n_bytes), 16)
        self.send = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.recv = torch.empty(n, dtype=torch.uint8, device=self.device)

    def _post(self, user, sends, n_sends, recvs, n_recvs, stream) -> int:
        try:
            import torch.distributed as dist

            # torch.distributed orders a point-to-point operation against torch's CURRENT stream (ProcessGroupNCCL records its
            # event there), not against the stream argument of this callback: they must be the same stream, or the gather
            # kernel that fills `send` is not ordered before the send
            if self.device != "cpu" and torch.cuda.is_available():
                cur = torch.cuda.current_stream().cuda_stream
                if int(stream or 0) != int(cur or 0):
                    raise RuntimeError(f"halo post on stream {stream}, but torch's current stream is {cur}: run the step "
                                       "under `torch.cuda.stream(s)` of the stream it launches on")
            self.exchanges += 1
            sh = self.shard
            to_global = (lambda r: dist.get_global_rank(sh.group, r)) if sh.group is not None else (lambda r: r)
            cut = lambda buf, m: buf[m.offset:m.offset + m.bytes]  # noqa: E731
            out = [(sends[i].peer, cut(self.send, sends[i])) for i in range(n_sends)]
            inc = [(recvs[i].peer, cut(self.recv, recvs[i])) for i in range(n_recvs)]
            if dist.get_backend(sh.group) == "gloo":   # tests: staged through host memory, synchronously
                host = [(peer, t, torch.empty(t.shape, dtype=t.dtype)) for peer, t in inc]
                ops = [dist.P2POp(dist.isend, t.cpu(), to_global(peer), sh.group) for peer, t in out]
                ops += [dist.P2POp(dist.irecv, h, to_global(peer), sh.group) for peer, _, h in host]
                for work in dist.batch_isend_irecv(ops):
                    work.wait()
                for _, t, h in host:
                    t.copy_(h)
                self.works = []
            else:
                ops = [dist.P2POp(dist.isend, t, to_global(peer), sh.group) for peer, t in out]
                ops += [dist.P2POp(dist.irecv, t, to_global(peer), sh.group) for peer, t in inc]
                self.works = dist.batch_isend_irecv(ops)
            return 0
        except BaseException as e:  # noqa: BLE001  (an exception must not unwind through the C frames)
            self.error = e
            return -1

    def _wait(self, user, stream) -> int:
        try:
            timed = self.time_waits and self.works
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for work in self.works:
                work.wait()   # stream-ordered: the current HIP stream waits, not the host
            if timed:
                e1.record()
                self._wait_events.append((e0, e1))
            self.works = []
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = e
            return -1


    def wait_ms(self) -> float:
        """Milliseconds the launch stream stood still in `wait` since the last call (synchronises; `time_waits` must be on)."""
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self._wait_events)
        self._wait_events = []
        return ms

    def selftest(self, n_bytes: int = 1 << 20) -> None:
        """Every rank sends a rank-stamped pattern to its neighbours through this very transport (`_post` / `_wait`, the
        staging buffers, the process group) and checks what arrived: a mis-wired fabric, rank order or stream order shows
        up here, before the first step, instead of as a wrong forecast.  Raises on failure."""
        sh = self.shard
        if sh is None or sh.world <= 1:
            return
        if self.send is None or self.send.numel() < 32:
            self.allocate(max(2 * n_bytes, 32))
        n = min(n_bytes, self.send.numel() // 2) // 16 * 16
        peers = [p for p in (sh.rank - 1, sh.rank + 1) if 0 <= p < sh.world]
        idx = torch.arange(n, device=self.device, dtype=torch.int64)
        msgs = (lib.HipHaloMsg * 2)()
        for i, peer in enumerate(peers):
            self.send[i * n:(i + 1) * n] = ((sh.rank * 31 + peer * 7 + idx) % 251).to(torch.uint8)
            msgs[i].peer, msgs[i].reserved, msgs[i].offset, msgs[i].bytes = peer, 0, i * n, n
        self.recv[:len(peers) * n] = 0xEE
        stream = lib._stream() if self.device != "cpu" else 0
        if self._post(None, msgs, len(peers), msgs, len(peers), stream) != 0 or self._wait(None, stream) != 0:
            raise RuntimeError(f"halo transport self-test failed on rank {sh.rank}") from self.error
        for i, peer in enumerate(peers):
            want = ((peer * 31 + sh.rank * 7 + idx) % 251).to(torch.uint8)
            if not torch.equal(self.recv[i * n:(i + 1) * n], want):
                bad = int((self.recv[i * n:(i + 1) * n] != want).nonzero()[0])
                raise RuntimeError(f"halo transport self-test: rank {sh.rank} received a wrong byte {bad} from rank {peer}")


class NativeModel:
    def __init__(self, model, shard=None, transport=None) -> None:
        L = lib.load()
        cfg = model.config
        if len(cfg.encoder_depths) > 4 or len(cfg.encoder_depths) != len(cfg.decoder_depths):
            raise NotImplementedError("the HIP engine runs U-nets of up to four stages, as deep on the way up as down")
        if cfg.lora_mode not in _LORA_MODES:
            raise ValueError(f"Invalid mode: {cfg.lora_mode}")
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
        # -- variant keywords --
        c.variant = _VARIANTS[model.variant]
        self.surf_inputs = wave_sources(model) if model.variant == "wave" else tuple(cfg.surf_vars)
        lc = tuple(cfg.level_condition or ())
        diff = type(model)._predict_difference_history_dim_lookup if model.variant == "air_pollution" else {}
        keep = {
            "surf": _strs(cfg.surf_vars), "static": _strs(cfg.static_vars), "atmos": _strs(cfg.atmos_vars),
            "lc": _dbl(lc), "sep": _strs(cfg.separate_perceiver), "mod": _strs(cfg.modulation_heads),
            "diff": _i32([diff.get(v, -1) for v in cfg.modulation_heads]),
            "pos_s": _strs(cfg.positive_surf_vars), "pos_a": _strs(cfg.positive_atmos_vars),
            "inputs": _strs(self.surf_inputs if model.variant == "wave" else ()),
            "dens": _strs(getattr(model, "density_channel_surf_vars", ())), "ang": _strs(getattr(model, "angle_surf_vars", ())),
        }
        self._keep_cfg = keep   # the handle copies what it needs in aurora_hip_create; kept for the call's duration
        c.n_surf, c.n_static, c.n_atmos = len(cfg.surf_vars), len(cfg.static_vars), len(cfg.atmos_vars)
        c.surf_vars, c.static_vars, c.atmos_vars = keep["surf"], keep["static"], keep["atmos"]
        c.n_level_condition, c.level_condition = len(lc), keep["lc"]
        c.dynamic_vars, c.atmos_static_vars = int(cfg.dynamic_vars), int(cfg.atmos_static_vars)
        c.clamp_at_first_step, c.simulate_indexing_bug = int(cfg.clamp_at_first_step), int(cfg.simulate_indexing_bug)
        c.n_separate_perceiver, c.separate_perceiver = len(cfg.separate_perceiver), keep["sep"]
        c.n_modulation_heads, c.modulation_heads, c.difference_history = len(cfg.modulation_heads), keep["mod"], keep["diff"]
        c.n_positive_surf, c.positive_surf_vars = len(cfg.positive_surf_vars), keep["pos_s"]
        c.n_positive_atmos, c.positive_atmos_vars = len(cfg.positive_atmos_vars), keep["pos_a"]
        if model.variant == "wave":
            c.n_surf_inputs, c.surf_inputs = len(self.surf_inputs), keep["inputs"]
            c.n_density, c.density_channel_surf_vars = len(model.density_channel_surf_vars), keep["dens"]
            c.n_angle, c.angle_surf_vars = len(model.angle_surf_vars), keep["ang"]
        c.tuning = lib.tuning_from_env()
        handle = ctypes.c_void_p()
        lib._check(L.aurora_hip_create(ctypes.byref(c), ctypes.byref(handle)))
        self._h = handle
        self.param_dtype = torch.float32
        for name, t in model.state_dict().items():
            t = t.detach()
            if t.dtype == torch.float64:
                # `model.double()` (the reference's golden test runs so, tests/test_model.py:18-24 upstream; docs/usage.md
                # callers too): the engine computes in fp32 (bf16 backbone GEMMs under autocast) -- the fp64 masters are
                # rounded once here, the predictions are returned as float64 like the reference's.  Said out loud, once.
                if self.param_dtype != torch.float64:
                    import warnings

                    warnings.warn("aurora_amd: the model's parameters are float64; the HIP engine computes in float32 "
                                  "(fp32 masters packed from them, predictions cast back to float64). Results match an "
                                  "fp64 evaluation to fp32 round-off (~1e-6 relative), not to fp64's.", UserWarning, stacklevel=3)
                self.param_dtype = torch.float64
                t = t.float()
            t = t.contiguous()
            if t.dtype != torch.float32:
                raise TypeError(f"aurora_amd: parameter '{name}' is {t.dtype}; the HIP engine packs float32 (or float64, "
                                "rounded to float32) masters -- keep the model in float32 and use `autocast=True` for bf16")
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            lib._check(L.aurora_hip_pack_weights(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim(),
                                                 lib.F32, int(t.is_cuda)))
        lib._check(L.aurora_hip_finalize(self._h, lib._stream()))
        n_out = L.aurora_hip_output_vars(self._h, None, 0)
        names = (ctypes.c_char_p * max(n_out, 1))()
        L.aurora_hip_output_vars(self._h, names, n_out)
        self.surf_outputs = tuple(names[i].decode() for i in range(n_out))
        # -- latitude band of a sharded forecast --
        self.shard = shard if (shard is not None and shard.world > 1) else None
        self.transport = None
        if self.shard is not None:
            device = next(model.parameters()).device
            self.transport = transport if transport is not None else _Transport(self.shard, device)
            band = lib.HipBand(self.shard.rank, self.shard.world, self.transport.post_c, self.transport.wait_c, None)
            lib._check(L.aurora_hip_set_band(self._h, ctypes.byref(band)))
        self.band_rows = None          # data rows [row0, row1) of the full grid this rank owns (sharded)
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
    def precompute(self, lat: torch.Tensor, lon: torch.Tensor, levels: tuple) -> None:
        """`lat` / `lon`: coordinates of the WHOLE (cropped) grid, vectors or matrices."""
        # storage identity first (no device->host copy per step), then content
        ident = (lat.data_ptr(), tuple(lat.shape), _version(lat), lon.data_ptr(), tuple(lon.shape), _version(lon), levels,
                 tuple(sorted(self.model.surf_stats.items())))
        if ident == self._grid_ident:
            return
        lat_h, lon_h = lat.detach().cpu(), lon.detach().cpu()
        key = (lat_h.numpy().tobytes(), lon_h.numpy().tobytes(), tuple(lat_h.shape), tuple(lon_h.shape), ident[6], ident[7])
        if key != self._grid_key:
            cfg, L = self.cfg, lib.load()
            H, W = lat_h.shape[0], lon_h.shape[-1]
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
            s_aff = [normalisation.surf_affine(n, self.model.surf_stats) for n in self.surf_inputs]
            t_aff = [normalisation.surf_affine(n, self.model.surf_stats) for n in cfg.static_vars]
            a_aff = [normalisation.atmos_affine(n, levels) for n in cfg.atmos_vars]
            keep = [lv, _dbl([a[0] for a in s_aff]), _dbl([a[1] for a in s_aff]), _dbl([a[0] for a in t_aff]),
                    _dbl([a[1] for a in t_aff]), _dbl([x for a in a_aff for x in a[0]]),
                    _dbl([x for a in a_aff for x in a[1]])]
            g.surf_loc, g.surf_scale, g.static_loc, g.static_scale, g.atmos_loc, g.atmos_scale = keep[1:]
            g.pos_encoding = pos.ctypes.data_as(lib._PF)
            g.scale_encoding = scale.ctypes.data_as(lib._PF)
            lib._check(L.aurora_hip_precompute(self._h, ctypes.byref(g), lib._stream()))
            if self.shard is not None:
                r0, r1 = ctypes.c_int32(), ctypes.c_int32()
                lib._check(L.aurora_hip_band_rows(self._h, ctypes.byref(r0), ctypes.byref(r1)))
                self.band_rows = (r0.value, r1.value)
                need = L.aurora_hip_band_staging_bytes(self._h)
                if self.transport.send is None or self.transport.send.numel() < need:
                    self.transport.allocate(need)
                lib._check(L.aurora_hip_set_band_staging(self._h, self.transport.send.data_ptr(), self.transport.recv.data_ptr(),
                                                         self.transport.send.numel()))
            self._grid_key = key
        self._grid_ident = ident
        self._keep_coords = (lat, lon)

    def full_lat(self):
        """Latitudes of the whole grid last given to `precompute` (a sharded rank keeps them: its band is a slice)."""
        return self._keep_coords[0] if getattr(self, "_keep_coords", None) is not None else None

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

    def profile_end_list(self) -> list:
        """[(kernel, ms, work)] per launch, in launch order (synchronises the device)."""
        n = ctypes.c_int(0)
        lib._check(lib.load().aurora_hip_profile_end_list(self._h, None, 0, ctypes.byref(n)))
        out = (lib.HipProfileEntry * max(n.value, 1))()
        lib._check(lib.load().aurora_hip_profile_end_list(self._h, out, n.value, ctypes.byref(n)))
        return [(e.kernel.decode(), e.ms, e.work) for e in out[:n.value]]

    def set_time(self, times) -> None:
        """Clock-dependent inputs: the absolute-time encoding (encoder.py:359-363) and the calendar fields of the dynamic
        variables (encoder.py:226-246), exactly the fields the reference reads off the datetime objects."""
        stamps = _dbl([t.timestamp() / 3600 for t in times])
        cal = _i32([x for t in times for x in (t.hour, t.weekday(), t.day)])
        lib._check(lib.load().aurora_hip_set_time_ex(self._h, stamps, cal, len(times), lib._stream()))

    def workspace_bytes(self) -> int:
        return int(lib.load().aurora_hip_workspace_bytes(self._h))

    def guard_words(self) -> tuple[float, float, float, float]:
        """The device-side range words of the last step (include/aurora_hip.h: aurora_hip_guard_words): max |encoder
        context| (0 inside the guarded chain), max |normalised atmospheric / surface input|, max |decoder context|."""
        out = (ctypes.c_float * 4)()
        lib._check(lib.load().aurora_hip_guard_words(self._h, out, lib._stream()))
        return tuple(float(v) for v in out)

    def generation(self) -> int:
        """Changes whenever the handle re-allocates device memory a captured hipGraph may point at."""
        return int(lib.load().aurora_hip_generation(self._h))

    # -- the step ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, batch: Batch, upload_time: bool = True, out=None) -> Batch:
        """`batch`: float32, on the device; the whole cropped grid, or this rank's `BandBatch` of a sharded forecast (the
        grid must then have been given to `precompute`).  `out`: see `Aurora.forward`."""
        cfg = self.cfg
        md = batch.metadata
        levels = tuple(md.atmos_levels)
        H, W = batch.spatial_shape
        first = next(iter(batch.surf_vars.values()))
        B, T = first.shape[:2]
        dev = first.device
        assert md.lat.shape[0] == H and md.lon.shape[-1] == W
        assert md.lat.dtype in (torch.float32, torch.float64), f"Latitude num. unstable: {md.lat.dtype}."
        assert md.lon.dtype in (torch.float32, torch.float64), f"Longitude num. unstable: {md.lon.dtype}."
        if self.shard is None:
            self.precompute(md.lat, md.lon, levels)
        else:
            assert self.band_rows is not None and H == self.band_rows[1] - self.band_rows[0], "not this rank's latitude band"
        if upload_time:
            self.set_time(md.time)

        def unknown(d_, known, what):
            extra = [k for k in d_ if k not in known]
            if extra:
                raise KeyError(f"{what} variable(s) {extra} are not variables of this model")

        unknown(batch.surf_vars, self.surf_inputs, "surface-level")
        unknown(batch.static_vars, cfg.static_vars, "static")
        unknown(batch.atmos_vars, cfg.atmos_vars, "atmospheric")

        def same_strides(ts):
            ts = [None if t is None else (t if t.dtype == torch.float32 else t.float()) for t in ts]
            real = [t for t in ts if t is not None]
            if len({t.stride() for t in real}) > 1 or any(t.stride(-1) != 1 for t in real):
                ts = [None if t is None else t.contiguous() for t in ts]
            return ts

        surf = same_strides([batch.surf_vars.get(n) for n in self.surf_inputs])
        stat = same_strides([batch.static_vars.get(n) for n in cfg.static_vars])
        atmos = same_strides([batch.atmos_vars.get(n) for n in cfg.atmos_vars])

        def dest(kind, name, shape):
            t = None if out is None else out[kind].get(name)
            if (t is not None and tuple(t.shape) == shape and t.dtype == torch.float32 and t.is_contiguous()
                    and t.device == dev):
                return t
            return torch.empty(shape, dtype=torch.float32, device=dev)

        have_s, have_a = set(batch.surf_vars), set(batch.atmos_vars)
        out_s = [dest(0, n, (B, 1, H, W)) if n in have_s else None for n in self.surf_outputs]
        out_a = [dest(1, n, (B, 1, len(levels), H, W)) if n in have_a else None for n in cfg.atmos_vars]
        ptrs = lambda ts: (ctypes.c_void_p * max(len(ts), 1))(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
        stride_of = lambda ts: next(t.stride() for t in ts if t is not None)  # noqa: E731
        io = lib.HipStepIO()
        io.B, io.T, io.rollout_step = B, T, md.rollout_step
        keep = [ptrs(surf), ptrs(stat), ptrs(atmos), ptrs(out_s), ptrs(out_a)]
        io.surf, io.stat, io.atmos, io.out_surf, io.out_atmos = keep
        io.surf_strides[:] = stride_of(surf)
        if any(t is not None for t in stat):
            io.static_strides[:] = stride_of(stat)
        io.atmos_strides[:] = stride_of(atmos)
        code = lib.load().aurora_hip_step(self._h, ctypes.byref(io), lib._stream())
        if self.transport is not None and self.transport.error is not None:
            err, self.transport.error = self.transport.error, None
            raise err
        lib._check(code)
        self._keepalive = (surf, stat, atmos)     # inputs must outlive the enqueued kernels
        new_md = derive_metadata(md, lat=md.lat.to(torch.float32), lon=md.lon.to(torch.float32),
                                 time=tuple(t + cfg.timestep for t in md.time), rollout_step=md.rollout_step + 1)
        # the prediction lists the variables in the order the reference's does: the inputs' order, except that the
        # ocean-wave variant returns the directions after the other variables (aurora.py:914-932)
        if self.param_dtype != torch.float32:   # dtype follows the first parameter (aurora.py:277-281 upstream)
            out_s = [None if t is None else t.to(self.param_dtype) for t in out_s]
            out_a = [None if t is None else t.to(self.param_dtype) for t in out_a]
        o_s = dict(zip(self.surf_outputs, out_s))
        if self.model.variant == "wave":
            surf_out = {n: o_s[n] for n in self.surf_outputs if o_s[n] is not None}
        else:
            surf_out = {n: o_s[n] for n in batch.surf_vars}
        o_a = dict(zip(cfg.atmos_vars, out_a))
        atmos_out = {n: o_a[n] for n in batch.atmos_vars}
        if isinstance(batch, BandBatch):
            return BandBatch(surf_out, dict(batch.static_vars), atmos_out, new_md, full_patch_rows=batch.full_patch_rows,
                             band=batch.band, rank=batch.rank, world=batch.world)
        return Batch(surf_out, dict(batch.static_vars), atmos_out, new_md)
