"""The MI355X engine: packs an `Aurora` model's weights for the HIP library and runs the step.

`Engine.step(batch)` is what `Aurora.forward` executes.  It reproduces the data flow of the
reference's forward (aurora/model/aurora.py:265-392; encoder.py:198-366; swin3d.py:884-936,
440-509; decoder.py:168-276) as a sequence of libaurora_hip calls on the current HIP stream:

  encoder   patchify(+normalise) -> GEMM -> surface MLP/LN, per-level GEMMs -> Perceiver level
            aggregation (GEMM, small cross attention, GEMM, LN, MLP, LN) -> token assembly with
            the cached position / scale / time embeddings
  backbone  per Swin block: qkv GEMM -> window attention (gather/scatter through the host
            geometry tables) -> proj GEMM -> AdaLN + residual -> fc1 GEMM (GELU) -> fc2 GEMM ->
            AdaLN + residual; merge / split around the U-net stages
  decoder   head GEMM + unpatchify (surface), Perceiver level de-aggregation, head GEMM +
            unpatchify with clamp / un-normalise fused

Everything that does not depend on the input fields is computed once and cached on the device:
AdaLN modulation vectors (lead time is a model constant), Fourier position / scale tables per
grid, pressure-level embeddings per level set, LoRA-merged weight sets per roll-out phase.

Precision: parameters must be fp32.  With `autocast=False` everything runs in fp32 (fp32-input
MFMA, exact fp32 FMA chains).  With `autocast=True` the backbone GEMMs and attention take bf16
operands with fp32 accumulation while LayerNorm statistics, softmax and the residual stream stay
fp32 -- the semantics of the reference's `torch.autocast` region (aurora.py:327-343); encoder and
decoder stay fp32 like upstream.

torch is used here for device memory (torch.empty / views), layout plumbing at pack time
(cat / pad / transpose of weights) and streams -- no torch arithmetic on the per-step path.
"""

from __future__ import annotations

import contextlib
import dataclasses
import os
from collections import OrderedDict
from datetime import timedelta
from typing import Optional, Sequence

import numpy as np
import torch

from aurora_amd import normalisation
from aurora_amd.batch import BandBatch, Batch, Metadata, derive_metadata
from aurora_amd.engine import encodings, geometry, lib, native, partition
from aurora_amd.model.schema import DYNAMIC_VARS, LORA_ALPHA, LORA_RANK
from aurora_amd.normalisation import level_to_str

F32 = torch.float32
BF16 = torch.bfloat16


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclasses.dataclass
class Shard:
    """Latitude-band sharding of one forecast over `world` ranks (see engine/partition.py)."""

    rank: int
    world: int
    group: object = None          # torch.distributed process group (None: default group)
    gather_output: bool = True    # True: forward() returns the full fields on every rank


def fused_ln_fills(rows: int, cus: int) -> bool:
    """Does a launch of the row-owning linear + LayerNorm kernel (128 x 512 tiles, one per CU and round) fill its rounds to
    at least 85 %?  (The rule of csrc/model.hip: 2,025 tiles on 256 CUs do, a latitude band's 270 do not.)"""
    tiles = -(-rows // 128)
    return tiles >= 0.85 * (-(-tiles // cus) * cus)


_F16_SAFE = 16384.0   # activations below this may take the two-term fp16 operand split (fp16 overflows at 65504)


@dataclasses.dataclass
class Exchange:
    """One halo exchange: tensors to send to / receive from peer ranks (contiguous, on the device)."""

    sends: list
    recvs: list
    deferred: bool = False   # True: the data are needed only when the generator yields `Complete`


class Complete:
    """Yielded after a deferred `Exchange`: the received halo rows must be in place before resuming."""


class Engine:
    def __init__(self, model) -> None:
        lib.load()  # fail loudly if libaurora_hip.so is missing
        self.model = model
        self.cfg = model.config
        p = next(model.parameters())
        if p.dtype != F32:
            raise NotImplementedError(
                f"aurora_amd computes in fp32 (or bf16 backbone with autocast=True); parameters "
                f"are {p.dtype}. Keep the model in float32."
            )
        self.device = p.device
        self.bb_dtype = BF16 if model.autocast else F32
        cfg = self.cfg
        for heads, dim in zip(cfg.encoder_num_heads + cfg.decoder_num_heads,
                              cfg.stage_dims() + cfg.stage_dims()[::-1]):
            if dim != heads * 64:
                raise NotImplementedError(
                    f"the window-attention kernel is built for head_dim 64 (dim {dim}, {heads} heads)"
                )
        if int(np.prod(cfg.window_size)) > 144:
            raise NotImplementedError("windows of more than 144 tokens are not supported")
        self._sd = {k: v.detach() for k, v in model.state_dict().items()}
        self._param_stamp = self._stamp()
        self._grid_cache: dict = {}
        self._level_cache: dict = {}
        self._table_cache: dict = {}
        self._ws_cache: dict = {}
        self._embed_w_cache: dict = {}
        self._embed_extra: dict = {}
        self._stat_cache: dict = {}
        self._lora_sets: "OrderedDict[object, dict]" = OrderedDict()
        self.debug_hook = None  # optional callable(tag, tensor) at stage boundaries (tools/debug_blocks.py)
        self.shard: Optional[Shard] = getattr(model, "_shard", None)
        self._plan_cache: dict = {}
        self._time_bufs: dict = {}     # B -> persistent device buffers of the clock-dependent inputs
        self._pinned: dict = {}        # (shape, dtype) -> ring of pinned staging buffers for `_upload`
        self._capturing = False        # True while a hipGraph of the step is being captured
        # The ERA5 model family on one device: the whole step is sequenced by the C-ABI handle (csrc/model.hip); this
        # class then only converts Batches to raw pointers.  AURORA_NATIVE_STEP=0 keeps the Python sequencing below
        # (same kernels, same order; used by tests to compare the two).
        self.native = None
        if (native.supports(model) and (self.shard is None or self.shard.world == 1)
                and os.environ.get("AURORA_NATIVE_STEP", "1") != "0"):
            self.native = native.NativeModel(model)
            self.blocks = self._block_list()
            return
        self._pack_static()

    # ---------------------------------------------------------------------------------------
    # helpers
    # ---------------------------------------------------------------------------------------
    def _stamp(self) -> int:
        try:
            return sum(p._version for p in self.model.parameters())
        except RuntimeError:  # inference tensors carry no version counter
            return -1

    def is_stale(self) -> bool:
        return self._stamp() != self._param_stamp

    def empty(self, *shape: int, dtype=F32) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _dev(self, arr: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def _p(self, name: str) -> torch.Tensor:
        return self._sd[name].contiguous()

    def _to_bb(self, w: torch.Tensor) -> torch.Tensor:
        """A weight in the backbone compute dtype (bf16 copy via the HIP convert kernel)."""
        w = w.contiguous()
        if self.bb_dtype == F32:
            return w
        return lib.convert(w, torch.empty_like(w, dtype=BF16))

    def _linear_new(self, a, w, bias, n_out, **kw) -> torch.Tensor:
        out = self.empty(a.shape[0], n_out, dtype=a.dtype)
        return lib.linear(a, w, bias, out, **kw)

    # ---------------------------------------------------------------------------------------
    # packing (input independent)
    # ---------------------------------------------------------------------------------------
    def _pack_static(self) -> None:
        cfg, sd = self.cfg, self._sd
        D = cfg.embed_dim
        hours = cfg.timestep / timedelta(hours=1)
        lead = self._dev(encodings.lead_time(hours, D)[None])  # (1, D) fp32

        # -- AdaLN modulation of every block: one GEMM over the stacked modulation weights --
        t1 = self._linear_new(lead, self._p("backbone.time_mlp.0.weight"), self._p("backbone.time_mlp.0.bias"),
                              D, act=lib.ACT_SILU)
        silu_c = self._linear_new(t1, self._p("backbone.time_mlp.2.weight"), self._p("backbone.time_mlp.2.bias"),
                                  D, act=lib.ACT_SILU)  # SiLU(c): the only way c is ever used
        self.blocks = self._block_list()
        names = [f"{blk['prefix']}.{n}.ln_modulation.1" for blk in self.blocks for n in ("norm1", "norm2")]
        w_all = torch.cat([sd[f"{n}.weight"] for n in names], dim=0).contiguous()
        b_all = torch.cat([sd[f"{n}.bias"] for n in names], dim=0).contiguous()
        mod = self._linear_new(silu_c, w_all, b_all, w_all.shape[0])[0]
        off = 0
        for blk in self.blocks:
            dim = blk["dim"]
            for n in ("norm1", "norm2"):
                # chunk(2): shift first, then scale (film.py:48); scale_bias is 0 in every config.
                blk[f"{n}.shift"] = mod[off:off + dim]
                blk[f"{n}.gain"] = mod[off + dim:off + 2 * dim]
                off += 2 * dim

        # -- backbone weights in the compute dtype --
        for blk in self.blocks:
            pre = blk["prefix"]
            blk["fc1.w"], blk["fc1.b"] = self._to_bb(sd[f"{pre}.mlp.fc1.weight"]), self._p(f"{pre}.mlp.fc1.bias")
            blk["fc2.w"], blk["fc2.b"] = self._to_bb(sd[f"{pre}.mlp.fc2.weight"]), self._p(f"{pre}.mlp.fc2.bias")
            blk["qkv.b"], blk["proj.b"] = self._p(f"{pre}.attn.qkv.bias"), self._p(f"{pre}.attn.proj.bias")
        self._lora_sets["base"] = {
            blk["prefix"]: (self._to_bb(sd[f"{blk['prefix']}.attn.qkv.weight"]),
                            self._to_bb(sd[f"{blk['prefix']}.attn.proj.weight"]))
            for blk in self.blocks
        }
        self.merges, self.splits = {}, {}
        n_enc, n_dec = len(cfg.encoder_depths), len(cfg.decoder_depths)
        for i in range(n_enc - 1):
            pre = f"backbone.encoder_layers.{i}.downsample"
            self.merges[i] = dict(w=self._to_bb(sd[f"{pre}.reduction.weight"]),
                                  ln_w=self._p(f"{pre}.norm.weight"), ln_b=self._p(f"{pre}.norm.bias"))
        for i in range(n_dec - 1):
            pre = f"backbone.decoder_layers.{i}.upsample"
            self.splits[i] = dict(w1=self._to_bb(sd[f"{pre}.lin1.weight"]), w2=self._to_bb(sd[f"{pre}.lin2.weight"]),
                                  ln_w=self._p(f"{pre}.norm.weight"), ln_b=self._p(f"{pre}.norm.bias"))

        # -- encoder / decoder constants that depend on parameters only --
        self.lead_emb = self._linear_new(lead, self._p("encoder.lead_time_embed.weight"),
                                         self._p("encoder.lead_time_embed.bias"), D)  # (1, D)
        self.enc_layers = self._pack_resampler("encoder.level_agg", cfg.enc_depth, cfg.num_heads)
        latents = self._p("encoder.atmos_latents")
        l0 = self.enc_layers[0]
        q0 = self._linear_new(latents, l0["to_q"], None, l0["to_q"].shape[0])
        if "ln_q.w" in l0:
            lib.layernorm(q0, l0["ln_q.w"], l0["ln_q.b"], out_f32=q0)
        self.enc_latents, self.enc_q0 = latents, q0
        self.dec_layers = {"main": self._pack_resampler("decoder.level_decoder", cfg.dec_depth, cfg.num_heads)}
        if cfg.dec_separate_perceiver:
            self.dec_layers["alt"] = self._pack_resampler("decoder.level_decoder_alternate", cfg.dec_depth,
                                                          cfg.num_heads)
        torch.cuda.current_stream().synchronize()

    def _pack_resampler(self, prefix: str, depth: int, heads: int) -> list[dict]:
        layers = []
        for i in range(depth):
            p = f"{prefix}.layers.{i}"
            d = dict(to_q=self._p(f"{p}.0.to_q.weight"), to_kv=self._p(f"{p}.0.to_kv.weight"),
                     to_out=self._p(f"{p}.0.to_out.weight"),
                     fc1_w=self._p(f"{p}.1.net.0.weight"), fc1_b=self._p(f"{p}.1.net.0.bias"),
                     fc2_w=self._p(f"{p}.1.net.2.weight"), fc2_b=self._p(f"{p}.1.net.2.bias"),
                     ln1_w=self._p(f"{p}.2.weight"), ln1_b=self._p(f"{p}.2.bias"),
                     ln2_w=self._p(f"{p}.3.weight"), ln2_b=self._p(f"{p}.3.bias"))
            if f"{p}.0.ln_k.weight" in self._sd:
                d.update({"ln_k.w": self._p(f"{p}.0.ln_k.weight"), "ln_k.b": self._p(f"{p}.0.ln_k.bias"),
                          "ln_q.w": self._p(f"{p}.0.ln_q.weight"), "ln_q.b": self._p(f"{p}.0.ln_q.bias")})
            d["inner"] = d["to_q"].shape[0]
            d["head_dim"] = d["inner"] // heads
            # largest L1 row norm of the value projection: |v| <= v_l1 * max |context| (pack time, one sync)
            d["v_l1"] = max(float(d["to_kv"][d["inner"]:].abs().sum(dim=1).max().item()), 1e-6)
            # The two-term fp16 split scales weights by 2^6 and assumes |activation| < 65504: it is only used if every
            # weight of the layer stays below 1000 and the MLP's input bound sqrt(D) * max|LN gain| + max|LN bias|
            # (what a LayerNorm output can reach) and its hidden layer stay inside the range; else three bf16 terms.
            w_max = max(float(d[k].abs().max().item()) for k in ("to_kv", "to_out", "fc1_w", "fc2_w"))
            ln_bound = float(d["ln1_w"].abs().max().item()) * d["ln1_w"].numel() ** 0.5 + float(d["ln1_b"].abs().max().item())
            d["f16_ok"] = w_max < 1000.0 and ln_bound < _F16_SAFE
            if d["f16_ok"]:
                # the same weights in the fp16-pair layout (scaled by 2^6): two-term GEMMs spend no VALU work on them
                for k in ("to_kv", "to_out", "fc1_w", "fc2_w"):
                    if lib.presplit_ok(*d[k].shape):
                        d[k + ".s"] = lib.split_f16(d[k], scale=64.0)
            layers.append(d)
        return layers

    def _block_list(self) -> list[dict]:
        cfg = self.cfg
        dims = cfg.stage_dims()
        n_dec = len(cfg.decoder_depths)
        blocks = []
        for i, depth in enumerate(cfg.encoder_depths):
            for j in range(depth):
                blocks.append(dict(prefix=f"backbone.encoder_layers.{i}.blocks.{j}", dim=dims[i], stage=i,
                                   heads=cfg.encoder_num_heads[i], shifted=j % 2 == 1, part="enc", layer=i, j=j))
        for i, depth in enumerate(cfg.decoder_depths):
            s = n_dec - 1 - i
            for j in range(depth):
                blocks.append(dict(prefix=f"backbone.decoder_layers.{i}.blocks.{j}", dim=dims[s], stage=s,
                                   heads=cfg.decoder_num_heads[i], shifted=j % 2 == 1, part="dec", layer=i, j=j))
        return blocks

    # -- LoRA weight sets ---------------------------------------------------------------------
    def _lora_key(self, step: int):
        """Which merged weight set roll-out step `step` uses (reference lora.py:105-129)."""
        cfg = self.cfg
        if not cfg.use_lora or step >= cfg.lora_steps:
            return "base"
        if cfg.lora_mode == "single":
            return 0
        if cfg.lora_mode == "from_second":
            return "base" if step == 0 else 0
        if cfg.lora_mode == "all":
            return step
        raise ValueError(f"Invalid mode: {cfg.lora_mode}")

    def step_signature(self, step: int):
        """Everything about a step that depends on the roll-out step and is baked into a captured graph:
        the LoRA weight set and whether positive variables are clamped (aurora.py:368-388)."""
        cfg = self.cfg
        new_step = step + 1
        clamp = bool(cfg.positive_surf_vars or cfg.positive_atmos_vars) and (
            new_step >= 1 if cfg.clamp_at_first_step else new_step > 1)
        return (self._lora_key(step), clamp)

    def _attn_weights(self, step: int) -> dict:
        key = self._lora_key(step)
        if key not in self._lora_sets:
            sd, out = self._sd, {}
            scaling = LORA_ALPHA / LORA_RANK
            assert scaling == 1.0
            for blk in self.blocks:
                pre = blk["prefix"]
                pair = []
                for which in ("qkv", "proj"):
                    w = sd[f"{pre}.attn.{which}.weight"].contiguous()
                    a = sd[f"{pre}.attn.lora_{which}.loras.{key}.lora_A"]  # (r, in)
                    b = sd[f"{pre}.attn.lora_{which}.loras.{key}.lora_B"]  # (out, r)
                    # W' = W + B A: a rank-8 GEMM (K zero-padded to one 32-wide fp32 K-tile) with the
                    # base weight as residual -- LoRA costs nothing per step afterwards.
                    a_t = torch.zeros((a.shape[1], 32), dtype=F32, device=self.device)
                    a_t[:, :LORA_RANK] = a.t()
                    b_p = torch.zeros((b.shape[0], 32), dtype=F32, device=self.device)
                    b_p[:, :LORA_RANK] = b
                    merged = lib.linear(b_p, a_t, None, torch.empty_like(w), residual=w)
                    pair.append(self._to_bb(merged))
                out[pre] = tuple(pair)
            self._lora_sets[key] = out
            while len(self._lora_sets) > 4:  # "all" mode: keep base + the most recent sets
                for k in self._lora_sets:
                    if k != "base" and k != key:
                        del self._lora_sets[k]
                        break
        return self._lora_sets[key]

    # -- per grid / per level-set constants -----------------------------------------------------
    def _grid(self, lat: torch.Tensor, lon: torch.Tensor) -> torch.Tensor:
        """pos_embed(pos) + scale_embed(scale) of the patch grid, (L, D) fp32, cached.

        Looked up by tensor identity first (a roll-out passes the same lat/lon objects from step
        to step, so no device->host copy happens per step), then by content.
        """
        # (storage address + shape, not object identity: `Batch.crop` makes a fresh view of the same coordinates
        # on every call, and a miss costs a device->host copy, i.e. a full synchronisation per step)
        ident = (lat.data_ptr(), tuple(lat.shape), native._version(lat), lon.data_ptr(), tuple(lon.shape), native._version(lon))
        hit = self._grid_cache.get("ident")
        if hit is not None and hit[0] == ident:
            return hit[3]
        lat_h, lon_h = lat.detach().cpu(), lon.detach().cpu()
        key = (tuple(lat_h.shape), tuple(lon_h.shape), lat_h.numpy().tobytes(), lon_h.numpy().tobytes())
        if key not in self._grid_cache:
            D, P = self.cfg.embed_dim, self.cfg.patch_size
            pos, scale = encodings.pos_scale_encodings(D, lat_h, lon_h, P)
            pe = self._linear_new(self._dev(pos), self._p("encoder.pos_embed.weight"),
                                  self._p("encoder.pos_embed.bias"), D)
            ps = lib.linear(self._dev(scale), self._p("encoder.scale_embed.weight"),
                            self._p("encoder.scale_embed.bias"), torch.empty_like(pe), residual=pe)
            if len(self._grid_cache) > 8:
                self._grid_cache.clear()
            self._grid_cache[key] = ps
        self._grid_cache["ident"] = (ident, lat, lon, self._grid_cache[key])  # holds lat/lon alive
        return self._grid_cache[key]

    def _levels(self, levels: tuple) -> dict:
        if levels not in self._level_cache:
            cfg = self.cfg
            D, D2 = cfg.embed_dim, 2 * cfg.embed_dim
            C = len(levels)
            enc = self._dev(encodings.levels(levels, D))
            # per-level bias of the atmospheric patch embedding: patch bias + level embedding
            if cfg.level_condition:
                pb = torch.stack([self._p(f"encoder.atmos_token_embeds.layers.{level_to_str(lv)}.bias")
                                  for lv in levels])
                bias = lib.linear(enc, self._p("encoder.atmos_levels_embed.weight"),
                                  self._p("encoder.atmos_levels_embed.bias"), self.empty(C, D), residual=pb)
            else:
                pb = self._p("encoder.atmos_token_embeds.bias")[None].expand(C, D)
                bias = lib.linear(enc, self._p("encoder.atmos_levels_embed.weight"),
                                  self._p("encoder.atmos_levels_embed.bias"), self.empty(C, D), residual=pb)
            dec = self._dev(encodings.levels(levels, D2))
            queries = self._linear_new(dec, self._p("decoder.atmos_levels_embed.weight"),
                                       self._p("decoder.atmos_levels_embed.bias"), D2)
            out = dict(enc_bias=bias, dec_queries=queries, enc_bias_max=float(bias.abs().max().item()))
            for name, layers in self.dec_layers.items():
                out[f"dec_q.{name}"] = self._linear_new(queries, layers[0]["to_q"], None, layers[0]["inner"])
            self._level_cache[levels] = out
        return self._level_cache[levels]

    def _tables(self, res, shifted: bool):
        key = (res, shifted)
        if key not in self._table_cache:
            tok, grp, _ = geometry.window_tables(tuple(res), tuple(self.cfg.window_size), shifted)
            self._table_cache[key] = (self._dev(tok), None if grp is None else self._dev(grp))
        return self._table_cache[key]

    def _stats(self, kind: str, name: str, levels: tuple):
        """Device (loc, scale, 1/scale) vectors of a variable, refreshed when the tables change."""
        if kind == "surf":
            loc, sc = normalisation.surf_affine(name, self.model.surf_stats)
            locs, scs = [loc], [sc]
        elif kind == "one":  # constant planes (dynamic variables): identity normalisation
            locs, scs = [0.0], [1.0]
        else:
            locs, scs = normalisation.atmos_affine(name, levels)
        key = (kind, name, levels)
        val = (tuple(locs), tuple(scs))
        hit = self._stat_cache.get(key)
        if hit is None or hit[0] != val:
            loc_t = torch.tensor(locs, dtype=F32, device=self.device)
            sc_t = torch.tensor(scs, dtype=F32, device=self.device)
            inv_t = torch.tensor([1.0 / s for s in scs], dtype=torch.float64).to(F32).to(self.device)
            hit = (val, loc_t, sc_t, inv_t)
            self._stat_cache[key] = hit
        return hit[1], hit[2], hit[3]

    def _surf_guard(self):
        """Constants of the surface MLP's guarded two-term chain (None if its weights rule the split out): pre-split
        weights, largest L1 row norm and |bias| of the first linear, max |embedding bias| + max |level encoding|."""
        if not hasattr(self, "_surf_guard_cache"):
            w0, w2 = self._p("encoder.surf_mlp.net.0.weight"), self._p("encoder.surf_mlp.net.2.weight")
            ok = (float(w0.abs().max().item()) < 1000.0 and float(w2.abs().max().item()) < 1000.0
                  and lib.presplit_ok(*w0.shape) and lib.presplit_ok(*w2.shape))
            self._surf_guard_cache = None if not ok else dict(
                w0_s=lib.split_f16(w0, scale=64.0), w2_s=lib.split_f16(w2, scale=64.0),
                l1_0=max(float(w0.abs().sum(dim=1).max().item()), 1e-6),
                b0=float(self._p("encoder.surf_mlp.net.0.bias").abs().max().item()),
                c=float(self._p("encoder.surf_token_embeds.bias").abs().max().item())
                + float(self._p("encoder.surf_level_encoding").abs().max().item()))
        return self._surf_guard_cache

    def _embed_weight(self, prefix: str, names: tuple, T: int) -> tuple[torch.Tensor, int]:
        """(D, Kpad) GEMM weight of a LevelPatchEmbed for the given variable order / history."""
        key = (prefix, names, T)
        if key not in self._embed_w_cache:
            ws = [self._sd[f"{prefix}.weights.{n}"][:, 0, :T] for n in names]  # (D, T, P, P) each
            w = torch.stack(ws, dim=1).reshape(ws[0].shape[0], -1)  # (D, V*T*P*P), (v, t, i, j) order
            K = w.shape[1]
            Kpad = _round_up(K, 32)
            wp = torch.zeros((w.shape[0], Kpad), dtype=F32, device=self.device)
            wp[:, :K] = w
            self._embed_w_cache[key] = (wp, K)
            # for the guarded two-term chain of the atmospheric embedding: largest L1 row norm (|embedding| <= l1 max|input| +
            # |bias|) and the weight in the fp16-pair layout, if eligible
            l1 = max(float(wp.abs().sum(dim=1).max().item()), 1e-6)
            ok = float(wp.abs().max().item()) < 1000.0 and lib.presplit_ok(*wp.shape)
            self._embed_extra[key] = (lib.split_f16(wp, scale=64.0) if ok else None, l1)
        return self._embed_w_cache[key]

    # ---------------------------------------------------------------------------------------
    # the step
    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, batch: Batch, out=None) -> Batch:
        """One forecast step.  With sharding, halo exchanges run as NCCL/RCCL point-to-point groups.
        `out`: optional destination tensors of the prediction, see `Aurora.forward`."""
        if self.native is not None:
            done = self._native_step(batch, out)
            if done is not None:
                return done
        gen = self.step_gen(batch, out)
        pending = []
        try:
            req = next(gen)
            while True:
                if isinstance(req, Complete):
                    for work in pending:
                        work.wait()  # stream-ordered: the current HIP stream waits, not the host
                    pending = []
                else:
                    pending = self._exchange(req)
                req = gen.send(None)
        except StopIteration as done:
            return done.value

    # -- per-launch timing (bench.py, tools): HIP events on the launch stream -----------------------------------
    def profile_start(self, only=None) -> None:
        """Bracket every launch (or the kernels named in `only`) with HIP events until `profile_stop`."""
        if self.native is not None:
            self.native.profile_begin(only)
        lib.profile_start(only)

    def profile_stop(self) -> dict:
        """{kernel: {"launches", "ms", "work"}}: launches issued by the C-ABI handle plus those issued from Python."""
        out = lib.profile_stop()
        if self.native is not None:
            for k, v in self.native.profile_end().items():
                d = out.setdefault(k, {"launches": 0, "ms": 0.0, "work": 0.0})
                for f in d:
                    d[f] += v[f]
        return out

    def _native_step(self, batch: Batch, out=None) -> Optional[Batch]:
        """The step through the C-ABI handle; None if this batch needs the Python sequencing (a variable subset, or
        latitude / longitude matrices)."""
        if isinstance(batch, BandBatch) or not self.native.accepts(batch):
            if not hasattr(self, "enc_layers"):
                self._pack_static()
            return None
        cfg = self.cfg
        batch = batch.type(F32).crop(cfg.patch_size).to(self.device)
        T = next(iter(batch.surf_vars.values())).shape[1]
        assert T <= cfg.max_history_size, f"{T} > {cfg.max_history_size}."
        return self.native.step(batch, upload_time=not self._capturing, out=out)

    def _exchange(self, req: Exchange) -> list:
        """Start one halo exchange with torch.distributed point-to-point operations; returns the works that are
        still in flight (to be waited for at the matching `Complete`; empty if the exchange was not deferred).

        backend "nccl" (= RCCL on ROCm): device tensors go straight over xGMI, grouped into one
        ncclGroup per exchange on RCCL's own stream, so kernels launched meanwhile (the interior windows) overlap
        the transfer; waiting makes the current HIP stream wait, not the host.
        backend "gloo" (tests: several processes sharing one GPU): staged through host memory, synchronously.
        """
        import torch.distributed as dist

        sh = self.shard
        to_global = (lambda r: dist.get_global_rank(sh.group, r)) if sh.group is not None else (lambda r: r)
        if dist.get_backend(sh.group) == "gloo":
            host_recv = [(peer, t, torch.empty(t.shape, dtype=t.dtype)) for peer, t in req.recvs]
            ops = [dist.P2POp(dist.isend, t.cpu(), to_global(peer), sh.group) for peer, t in req.sends]
            ops += [dist.P2POp(dist.irecv, h, to_global(peer), sh.group) for peer, _, h in host_recv]
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            for _, t, h in host_recv:
                t.copy_(h)
            return []
        ops = [dist.P2POp(dist.isend, t, to_global(peer), sh.group) for peer, t in req.sends]
        ops += [dist.P2POp(dist.irecv, t, to_global(peer), sh.group) for peer, t in req.recvs]
        works = dist.batch_isend_irecv(ops)
        if req.deferred:
            return works
        for work in works:
            work.wait()
        return []

    def step_gen(self, batch: Batch, out=None):
        """Generator form of the step: yields `Exchange` requests (sharded mode only) and returns the
        prediction.  `step()` drives it with RCCL; tests drive several ranks in one process."""
        model, cfg = self.model, self.cfg
        if not self._capturing:  # (a captured step runs on a batch the hook has already seen)
            batch = model.batch_transform_hook(batch)
        P, D = cfg.patch_size, cfg.embed_dim
        sh = self.shard if (self.shard is not None and self.shard.world > 1) else None
        band = None
        if sh is not None and isinstance(batch, BandBatch):
            band, full_rows = tuple(batch.band), batch.full_patch_rows
            batch = batch.type(F32).to(self.device)
        else:
            batch = batch.type(F32).crop(cfg.patch_size).to(self.device)
            full_rows = batch.spatial_shape[0] // P
        md = batch.metadata
        levels = tuple(md.atmos_levels)
        B, T = next(iter(batch.surf_vars.values())).shape[:2]
        W = batch.spatial_shape[1]
        Wp = W // P
        patch_res = (cfg.latent_levels, full_rows, Wp)
        n_enc = len(cfg.encoder_depths)
        all_res, _ = geometry.stage_resolutions(patch_res, n_enc)
        rows = None
        if sh is not None:
            assert B == 1, "latitude-band sharding runs one forecast (batch size 1) across the ranks"
            rows = partition.band_rows(all_res, tuple(cfg.window_size), sh.world)
            h0, h1 = rows[0][sh.rank]
            if band is None:  # full batch given: take this rank's band (views, no copy)
                cut = lambda d_: {k: v[..., h0 * P:h1 * P, :] for k, v in d_.items()}  # noqa: E731
                batch = BandBatch(cut(batch.surf_vars), cut(batch.static_vars), cut(batch.atmos_vars),
                                  derive_metadata(md, lat=md.lat[h0 * P:h1 * P]), full_patch_rows=full_rows,
                                  band=(h0, h1), rank=sh.rank, world=sh.world)
                md = batch.metadata
            else:
                assert band == (h0, h1), f"band {band} does not match this rank's rows {(h0, h1)}"
        H = batch.spatial_shape[0]
        Hp = H // P
        assert T <= cfg.max_history_size, f"{T} > {cfg.max_history_size}."
        assert md.lat.shape[0] == H and md.lon.shape[-1] == W
        assert md.lat.dtype in (torch.float32, torch.float64), f"Latitude num. unstable: {md.lat.dtype}."
        assert md.lon.dtype in (torch.float32, torch.float64), f"Longitude num. unstable: {md.lon.dtype}."
        assert cfg.latent_levels % cfg.window_size[0] == 0, "latent levels must be divisible by ws[0]"

        trace = os.environ.get("AURORA_TRACE")
        if trace:
            import sys
            import time

            def mark(tag, t0=[time.perf_counter()]):  # noqa: B006
                torch.cuda.synchronize()
                now = time.perf_counter()
                print(f"[aurora_amd] {tag}: {(now - t0[0]) * 1e3:.1f} ms", file=sys.stderr, flush=True)
                t0[0] = now
        else:
            mark = lambda tag: None  # noqa: E731
        mark("step start")
        self._cur_band = (full_rows, rows[0][sh.rank]) if sh is not None else None
        x_f, x_b = self._encode(batch, B, T, H, W, Hp, Wp, levels)
        mark("encoder")
        x_cat = yield from self._backbone(x_f, x_b, B, patch_res, md.rollout_step, rows)
        mark("backbone")
        pred = self._decode(x_cat, batch, B, H, W, Hp, Wp, levels, out)
        mark("decoder")
        if sh is not None and sh.gather_output:
            pred = self._gather(pred, rows[0], P)
        return pred

    def capture(self, batch: Batch) -> "GraphedStep":
        """Capture one step on (a private copy of) `batch` into a hipGraph; see GraphedStep."""
        return GraphedStep(self, batch)

    def local_band(self, batch: Batch) -> BandBatch:
        """This rank's latitude band of a full (cropped) batch, as views."""
        if isinstance(batch, BandBatch):
            return batch
        cfg, sh = self.cfg, self.shard
        P = cfg.patch_size
        H, W = batch.spatial_shape
        all_res, _ = geometry.stage_resolutions((cfg.latent_levels, H // P, W // P), len(cfg.encoder_depths))
        h0, h1 = partition.band_rows(all_res, tuple(cfg.window_size), sh.world)[0][sh.rank]
        cut = lambda d_: {k: v[..., h0 * P:h1 * P, :] for k, v in d_.items()}  # noqa: E731
        md = derive_metadata(batch.metadata, lat=batch.metadata.lat[h0 * P:h1 * P])
        return BandBatch(cut(batch.surf_vars), cut(batch.static_vars), cut(batch.atmos_vars), md,
                         full_patch_rows=H // P, band=(h0, h1), rank=sh.rank, world=sh.world)

    def _gather(self, pred: BandBatch, rows0, P: int) -> Batch:
        """Assemble the full fields on every rank: each rank broadcasts its band into place."""
        import torch.distributed as dist

        sh = self.shard
        H = rows0[-1][1] * P
        to_global = (lambda r: dist.get_global_rank(sh.group, r)) if sh.group is not None else (lambda r: r)
        staged = dist.get_backend(sh.group) == "gloo"

        def bcast(piece, src):
            if staged:
                h = piece.cpu()
                dist.broadcast(h, src=to_global(src), group=sh.group)
                piece.copy_(h)
            else:
                dist.broadcast(piece, src=to_global(src), group=sh.group)

        def gather(d_, lead):
            out = {}
            for k, v in d_.items():
                full = torch.empty((*v.shape[:-2], H, v.shape[-1]), dtype=v.dtype, device=v.device)
                for r, (a, b) in enumerate(rows0):
                    piece = v.contiguous() if r == sh.rank else torch.empty(
                        (*v.shape[:-2], (b - a) * P, v.shape[-1]), dtype=v.dtype, device=v.device)
                    bcast(piece, r)
                    full[..., a * P:b * P, :] = piece
                out[k] = full
            return out

        md = pred.metadata
        lat_parts = []
        for r, (a, b) in enumerate(rows0):
            piece = md.lat.contiguous() if r == sh.rank else torch.empty((b - a) * P, dtype=md.lat.dtype,
                                                                        device=md.lat.device)
            bcast(piece, r)
            lat_parts.append(piece)
        return Batch(gather(pred.surf_vars, 2), gather(pred.static_vars, 0), gather(pred.atmos_vars, 3),
                     derive_metadata(md, lat=torch.cat(lat_parts)))

    # -- clock-dependent inputs -----------------------------------------------------------------
    def _time_inputs(self, times, B: int) -> dict:
        """Device buffers holding everything the step derives from `metadata.time`: the absolute-time
        Fourier encoding (encoder.py:359-363) and, for dynamic-variable models, the six time-of-day /
        day-of-week / day-of-year planes (encoder.py:226-246).  They live in persistent buffers that are
        refreshed from the host BEFORE the step; a captured hipGraph only reads them."""
        if self.native is not None and not self._capturing and self.native._grid_key is not None:
            self.native.set_time(times)
        D = self.cfg.embed_dim
        bufs = self._time_bufs.get(B)
        if bufs is None:
            bufs = {"abs_enc": self.empty(B, D), "dyn": self.empty(6, B)}
            self._time_bufs[B] = bufs
        if not self._capturing:
            stamps = [t.timestamp() / 3600 for t in times]
            self._upload(bufs["abs_enc"], encodings.absolute_time(stamps, D))
            if self.cfg.dynamic_vars:
                vals = np.array([[np.cos(2 * np.pi * t.hour / 24), np.sin(2 * np.pi * t.hour / 24),
                                  np.cos(2 * np.pi * t.weekday() / 7), np.sin(2 * np.pi * t.weekday() / 7),
                                  np.cos(2 * np.pi * t.day / 365.25), np.sin(2 * np.pi * t.day / 365.25)]
                                 for t in times], dtype=np.float64).astype(np.float32)  # (B, 6)
                self._upload(bufs["dyn"], np.ascontiguousarray(vals.T))
        return bufs

    def _upload(self, dst: torch.Tensor, arr: np.ndarray) -> None:
        """Host -> device copy that does not stall the host: staged through a small ring of pinned buffers (a
        pageable `copy_` waits for everything queued before it, i.e. for the whole previous step)."""
        ring = self._pinned.setdefault((tuple(dst.shape), dst.dtype), {"slots": [], "next": 0})
        if len(ring["slots"]) < 4:
            ring["slots"].append((torch.empty(dst.shape, dtype=dst.dtype).pin_memory(), torch.cuda.Event()))
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"] % 4]
            slot[1].synchronize()          # the copy that used this slot four uploads ago has finished
        ring["next"] += 1
        slot[0].copy_(torch.from_numpy(arr).reshape(dst.shape))
        dst.copy_(slot[0], non_blocking=True)
        slot[1].record()

    # -- encoder ------------------------------------------------------------------------------
    def _var_desc(self, t: torch.Tensor, kind: str, name: str, levels: tuple, transform=0, comb=None) -> lib.PatchVar:
        loc, _, inv = self._stats(kind, name, levels)
        if kind == "surf" and t.dim() == 2:      # static (H, W)
            sb = st = sc = 0
            sh, sw = t.stride()
        elif kind == "surf":                      # (B, T, H, W)
            sb, st, sh, sw = t.stride()
            sc = 0
        elif kind == "one":                       # (B,) constant plane per batch element
            sb, st, sc, sh, sw = t.stride(0), 0, 0, 0, 0
        elif t.dim() == 2:                        # static fed at every level
            sb = st = sc = 0
            sh, sw = t.stride()
        elif t.dim() == 1:                        # constant plane fed at every level
            sb, st, sc, sh, sw = t.stride(0), 0, 0, 0, 0
        else:                                     # (B, T, C, H, W)
            sb, st, sc, sh, sw = t.stride()
        tw0 = tw1 = tb = 0.0
        if comb is not None:
            tw0, tw1, tb = comb
        return lib.PatchVar(t.data_ptr(), sb, st, sc, sh, sw, loc.data_ptr(), inv.data_ptr(), transform, tw0, tw1, tb)

    def _wave_channels(self, names: tuple) -> list:
        """Model input channels of the ocean-wave variant for the surface variables `names`:
        [(channel name, source variable, patchify transform code)], in the order the reference's
        `_pre_encoder_hook` leaves the dictionary (kept variables, then the appended channels)."""
        model = self.model
        kept, appended = [], []
        for k in names:
            dens = k in model.density_channel_surf_vars and f"{k}_density" not in names
            ang = k in model.angle_surf_vars and not (f"{k}_sin" in names and f"{k}_cos" in names)
            if not ang:
                kept.append((k, k, 4 if dens else 0))
            if dens:
                appended.append((f"{k}_density", k, 3))
            if ang:
                appended += [(f"{k}_sin", k, 5), (f"{k}_cos", k, 6)]
        return kept + appended

    def _combiner(self, kind: str, name: str):
        """(w0, w1, b) of the air-pollution Linear(2, 1) feature combiner, read to the host once."""
        key = ("combiner", kind, name)
        if key not in self._stat_cache:
            w = self._sd[f"{kind}_feature_combiner.{name}.weight"].reshape(-1).tolist()
            b = self._sd[f"{kind}_feature_combiner.{name}.bias"].reshape(-1).tolist()
            self._stat_cache[key] = (w[0], w[1], b[0])
        return self._stat_cache[key]

    def _encode(self, batch: Batch, B, T, H, W, Hp, Wp, levels):
        cfg, model = self.cfg, self.model
        P, D = cfg.patch_size, cfg.embed_dim
        L, C = Hp * Wp, len(levels)
        keep = []  # tensors whose storage must outlive the enqueued kernels of this call

        def transform_of(kind, name):
            pos = cfg.positive_surf_vars if kind == "surf" else cfg.positive_atmos_vars
            if name not in pos:
                return 0, None
            if model.variant == "air_pollution":
                return 2, self._combiner(kind, name)
            return 1, None

        f32c = lambda t: t if t.dtype == F32 else t.to(F32)  # noqa: E731
        surf = {k: f32c(v) for k, v in batch.surf_vars.items()}
        static = {k: f32c(v) for k, v in batch.static_vars.items()}
        atmos = {k: f32c(v) for k, v in batch.atmos_vars.items()}
        keep += list(surf.values()) + list(static.values()) + list(atmos.values())

        if model.variant == "wave":
            # density channels and sin/cos of directions (aurora.py:892-912) as per-channel transforms
            chans = self._wave_channels(tuple(surf))
            surf_names = tuple(n for n, _, _ in chans) + tuple(static)
            descs = [self._var_desc(surf[src], "surf", src, levels, code) for _, src, code in chans]
        else:
            surf_names = tuple(surf) + tuple(static)
            descs = [self._var_desc(v, "surf", k, levels, *transform_of("surf", k)) for k, v in surf.items()]
        descs += [self._var_desc(v, "surf", k, levels) for k, v in static.items()]
        tbufs = self._time_inputs(batch.metadata.time, B)
        dyn_t = []
        if cfg.dynamic_vars:
            dyn_t = [tbufs["dyn"][i] for i in range(6)]  # (B,) constant plane per batch element
            surf_names += DYNAMIC_VARS
            descs += [self._var_desc(t, "one", n, levels) for n, t in zip(DYNAMIC_VARS, dyn_t)]

        atmos_names = tuple(atmos)
        adescs = [self._var_desc(v, "atmos", k, levels, *transform_of("atmos", k)) for k, v in atmos.items()]
        if cfg.atmos_static_vars:
            if cfg.dynamic_vars:
                extra = list(static.items()) + list(zip(DYNAMIC_VARS, dyn_t))
                atmos_names += tuple(f"static_{n}" for n, _ in extra)
            else:  # the reference appends the bare static names here (encoder.py:268-269)
                extra = list(static.items())
                atmos_names += tuple(n for n, _ in extra)
            for n, t in extra:
                kind = "one" if t.dim() == 1 else "surf"
                d = self._var_desc(t, kind, n, levels)
                if kind == "surf":  # static plane normalised with its surface statistics, every level
                    loc, _, inv = self._stats("surf", n, levels)
                    loc_c, inv_c = loc.expand(C).contiguous(), inv.expand(C).contiguous()
                    keep += [loc_c, inv_c]
                    d.loc, d.inv_scale = loc_c.data_ptr(), inv_c.data_ptr()
                else:
                    loc, _, inv = self._stats("one", n, levels)
                    loc_c, inv_c = loc.expand(C).contiguous(), inv.expand(C).contiguous()
                    keep += [loc_c, inv_c]
                    d.loc, d.inv_scale = loc_c.data_ptr(), inv_c.data_ptr()
                adescs.append(d)
        if cfg.simulate_indexing_bug and "z" in atmos_names:
            # the slot of `static_z` is fed with `z`'s data (encoder.py:293-303)
            adescs[atmos_names.index("static_z")] = adescs[atmos_names.index("z")]

        # ---- surface level ----
        w_s, K_s = self._embed_weight("encoder.surf_token_embeds", surf_names, T)
        A_s = self.empty(B * L, w_s.shape[1])
        for i in range(0, len(descs), 32):
            lib.patchify(descs[i:i + 32], A_s, i * T * P * P, K_s, B, T, 1, Hp, Wp, P)
        sle = self._p("encoder.surf_level_encoding")[None].expand(B * L, D)  # stride-0 residual rows
        w0, b0 = self._p("encoder.surf_mlp.net.0.weight"), self._p("encoder.surf_mlp.net.0.bias")
        w2, b2 = self._p("encoder.surf_mlp.net.2.weight"), self._p("encoder.surf_mlp.net.2.bias")
        be = self._p("encoder.surf_token_embeds.bias")
        sg = self._surf_guard()
        w_s_s, l1e = self._embed_extra[("encoder.surf_token_embeds", surf_names, T)]
        if sg is not None and w_s_s is not None and lib.two_term_free():
            # Guarded like the atmospheric chain (csrc/model.hip): max |normalised input| once; every linear takes two fp16
            # terms iff the bound that word implies for ITS activation operand is inside fp16's range, else three bf16 terms.
            word = lib.absmax(A_s)
            lims = (_F16_SAFE, (_F16_SAFE - sg["c"]) / l1e, ((_F16_SAFE - sg["b0"]) / sg["l1_0"] - sg["c"]) / l1e)

            def pair(a, wf, ws, bias, n_out, limit, **kw):
                out = self.empty(a.shape[0], n_out)
                with lib.f32_gemm(2, guard=(word, limit)):
                    lib.linear(a, ws, bias, out, presplit=lib.F32_W_SPLIT, **kw)
                with lib.f32_gemm(1, guard=(word, limit)):
                    return lib.linear(a, wf, bias, out, **kw)

            xs0 = pair(A_s, w_s, w_s_s, be, D, lims[0], residual=sle)
            hid = pair(xs0, w0, sg["w0_s"], b0, w0.shape[0], lims[1], act=lib.ACT_GELU)
            y = pair(hid, w2, sg["w2_s"], b2, D, lims[2])
        else:
            xs0 = lib.linear(A_s, w_s, be, self.empty(B * L, D), residual=sle)
            hid = self._linear_new(xs0, w0, b0, w0.shape[0], act=lib.ACT_GELU)
            y = self._linear_new(hid, w2, b2, D)
        xs1 = self.empty(B * L, D)
        lib.layernorm(y, self._p("encoder.surf_norm.weight"), self._p("encoder.surf_norm.bias"), res=xs0, out_f32=xs1)
        del hid, y, A_s

        # ---- atmospheric levels ----
        lv = self._levels(levels)
        if cfg.level_condition:
            packs = [self._embed_weight(f"encoder.atmos_token_embeds.layers.{level_to_str(l_)}", atmos_names, T)
                     for l_ in levels]
        else:
            packs = [self._embed_weight("encoder.atmos_token_embeds", atmos_names, T)] * C
        K_a = packs[0][1]
        A_a = self.empty(C * B * L, packs[0][0].shape[1])
        for i in range(0, len(adescs), 32):
            lib.patchify(adescs[i:i + 32], A_a, i * T * P * P, K_a, B, T, C, Hp, Wp, P)
        xa = self.empty(C * B * L, D)
        R = B * L
        # The patch embedding and the level aggregation's to_kv as one guarded chain (as csrc/model.hip sequences it):
        # max |normalised input| is measured once; inside fp16's range -- together with the bound it implies for the
        # embeddings, |x| <= l1 * max|input| + max|bias| -- the embedding runs on two fp16 terms and writes fp16 PAIRS,
        # which to_kv multiplies without splitting anything; otherwise both run on three bf16 terms over fp32 buffers.
        if cfg.level_condition:
            ekeys = [(f"encoder.atmos_token_embeds.layers.{level_to_str(l_)}", atmos_names, T) for l_ in levels]
        else:
            ekeys = [("encoder.atmos_token_embeds", atmos_names, T)] * C
        extras = [self._embed_extra[k] for k in ekeys]
        chain = (lib.two_term_free() and all(e[0] is not None for e in extras)
                 and all(ly["f16_ok"] and "to_kv.s" in ly for ly in self.enc_layers))
        guard = None
        if chain:
            word = lib.absmax(A_a)
            l1, cb = max(e[1] for e in extras), lv["enc_bias_max"]
            guard = dict(word=word, a=l1, c=cb, limit_kv=min(_F16_SAFE, (_F16_SAFE - cb) / l1), pairs=True)
        for c in range(C):
            a_c, x_c = A_a[c * R:(c + 1) * R], xa[c * R:(c + 1) * R]
            if chain:
                with lib.f32_gemm(2, guard=(guard["word"], guard["limit_kv"])):
                    lib.linear(a_c, extras[c][0], lv["enc_bias"][c], x_c, presplit=lib.F32_W_SPLIT | lib.F32_C_SPLIT)
                with lib.f32_gemm(1, guard=(guard["word"], guard["limit_kv"])):
                    lib.linear(a_c, packs[c][0], lv["enc_bias"][c], x_c)
            else:
                lib.linear(a_c, packs[c][0], lv["enc_bias"][c], x_c)
        del A_a

        # ---- level aggregation (Perceiver resampler over the level axis) ----
        n_lat = cfg.latent_levels - 1
        lat = self._resampler(self.enc_layers, xa, q0=self.enc_q0, latents0=self.enc_latents, B=B, cols=L,
                              kv_bstride=L, kv_lstride=B * L, Lq=n_lat, Lk=C, heads=cfg.num_heads,
                              eps=cfg.perceiver_ln_eps, guard=guard)
        del xa

        # ---- assemble tokens + position / scale / time embeddings ----
        pos_scale = self._grid(batch.metadata.lat, batch.metadata.lon)
        time_emb = lib.linear(tbufs["abs_enc"], self._p("encoder.absolute_time_embed.weight"),
                              self._p("encoder.absolute_time_embed.bias"), self.empty(B, D),
                              residual=self.lead_emb.expand(B, D))
        Cl = cfg.latent_levels
        x_f = self.empty(B * Cl * L, D)
        x_b = self.empty(B * Cl * L, D, dtype=BF16) if self.bb_dtype == BF16 else None
        lib.assemble_tokens(xs1, lat, pos_scale, time_emb, x_f, x_b, B, Cl, L, D)
        self._keepalive = keep
        return x_f, x_b

    def _resampler(self, layers, ctx, *, q0, latents0, B, cols, kv_bstride, kv_lstride, Lq, Lk, heads, eps, guard=None):
        """PerceiverResampler (perceiver.py:212-233) for all grid columns at once.

        ctx: context rows, key j of column (b, l) at row b*kv_bstride + j*kv_lstride + l.
        First layer: the latents (and so q) are the same for every column.  Returns (B*cols*Lq, D).
        """
        lat = None
        n_rows = B * cols * Lq
        # The context is as unbounded as the model inputs (raw `randn` fields reach 4e6 here), so the linears that
        # read it, or averages of its value projection, pick their operand split on the device from max |ctx|.
        # ... from max |ctx|, measured here, or from the bound the caller derived from a word it measured upstream
        # (`guard`: max|ctx| <= a * word + c; `pairs`: ctx holds fp16 pairs iff word < limit_kv, fp32 otherwise)
        ctx_max = guard["word"] if guard else lib.absmax(ctx)
        g_a, g_c = (guard["a"], guard["c"]) if guard else (1.0, 0.0)
        ctx_pairs = bool(guard and guard["pairs"])
        for i, ly in enumerate(layers):
            inner, hd = ly["inner"], ly["head_dim"]
            bounded = lib.bounded_activations if ly["f16_ok"] else (lambda guard=None: contextlib.nullcontext())
            pre = ly["f16_ok"] and lib.two_term_free()

            def guarded(a, name, n_out, limit, a_pairs=False):
                """Guarded linear.  With pre-split weights: the two-term launch runs iff the guard holds, the three-term
                one on the fp32 weights iff it does not (include/aurora_hip.h) -- the same result as the guarded call."""
                if pre and name + ".s" in ly:
                    out = self.empty(a.shape[0], n_out)
                    with lib.f32_gemm(2, guard=(ctx_max, limit)):
                        lib.linear(a, ly[name + ".s"], None, out,
                                   presplit=lib.F32_W_SPLIT | (lib.F32_A_SPLIT if a_pairs else 0))
                    with lib.f32_gemm(1, guard=(ctx_max, limit)):
                        return lib.linear(a, ly[name], None, out)
                with bounded(guard=(ctx_max, limit)):
                    return self._linear_new(a, ly[name], None, n_out)

            assert not ctx_pairs or (pre and "to_kv.s" in ly)
            kv = guarded(ctx, "to_kv", 2 * inner, guard["limit_kv"] if ctx_pairs else (_F16_SAFE - g_c) / g_a, ctx_pairs)
            if "ln_k.w" in ly:  # LayerNorm over the K half, in place (perceiver.py:144-147)
                lib.layernorm(kv, ly["ln_k.w"], ly["ln_k.b"], out_f32=kv, d=inner)
            if i == 0:
                q, q_stride = q0, 0
            else:
                q = self._linear_new(lat, ly["to_q"], None, inner)
                if "ln_q.w" in ly:
                    lib.layernorm(q, ly["ln_q.w"], ly["ln_q.b"], out_f32=q)
                q_stride = Lq
            # |att| <= max |v| <= (largest L1 row norm of W_v) * max |ctx|: same guard, tighter limit.  With pre-split to_out
            # weights the attention writes fp16 pairs iff that guard holds, and to_out multiplies them without splitting.
            lim_out = (_F16_SAFE / ly["v_l1"] - g_c) / g_a
            att_pairs = pre and "to_out.s" in ly and inner % 32 == 0
            att = lib.perceiver_attention(q, q_stride, kv, self.empty(n_rows, inner), B, cols, kv_bstride,
                                          kv_lstride, Lq, Lk, heads, hd,
                                          pair_guard=(ctx_max, lim_out) if att_pairs else None)
            del kv
            D = ly["to_out"].shape[0]
            o = guarded(att, "to_out", D, lim_out, att_pairs)
            del att
            lat1 = self.empty(n_rows, D)   # fp32 values, or their fp16 pairs
            # The MLP in the fp16-pair layout end to end: LayerNorm writes its result already split (and only split),
            # fc1 reads that and writes its GELU'd result split, fc2 reads that -- neither GEMM splits anything -- and
            # the LayerNorm behind the MLP takes the split array as its residual.
            pairs = pre and "fc1_w.s" in ly and "fc2_w.s" in ly and D % 32 == 0
            res_kw = dict(res=latents0, res_mod=Lq) if i == 0 else dict(res=lat)
            if pairs:
                lib.layernorm(o, ly["ln1_w"], ly["ln1_b"], eps=eps, out_t=lat1, split_t=True, **res_kw)
            else:
                lib.layernorm(o, ly["ln1_w"], ly["ln1_b"], out_f32=lat1, eps=eps, **res_kw)
            del o
            if pairs:
                both = lib.F32_A_SPLIT | lib.F32_W_SPLIT
                hid = self.empty(n_rows, ly["fc1_w"].shape[0])
                lib.linear(lat1, ly["fc1_w.s"], ly["fc1_b"], hid, act=lib.ACT_GELU, presplit=both | lib.F32_C_SPLIT)
                y = lib.linear(hid, ly["fc2_w.s"], ly["fc2_b"], self.empty(n_rows, D), presplit=both)
            else:
                with bounded():    # fc1 sees a LayerNorm output (|x| <= sqrt(D) * gain), fc2 its GELU
                    hid = self._linear_new(lat1, ly["fc1_w"], ly["fc1_b"], ly["fc1_w"].shape[0], act=lib.ACT_GELU)
                    y = self._linear_new(hid, ly["fc2_w"], ly["fc2_b"], D)
            del hid
            lib.layernorm(y, ly["ln2_w"], ly["ln2_b"], res=lat1, out_f32=y, eps=eps, split_res=pairs)
            lat = y
        return lat

    # -- backbone -----------------------------------------------------------------------------
    def _plans(self, res, shifted: bool, rows_s):
        """Device copies of this rank's attention plan for one block flavour (None when un-sharded)."""
        key = (tuple(res), shifted, tuple(rows_s))
        if key not in self._plan_cache:
            p = partition.block_plans(tuple(res), tuple(self.cfg.window_size), shifted, tuple(rows_s))[self.shard.rank]
            d = dict(tok=self._dev(p.tok), grp=None if p.grp is None else self._dev(p.grp), n_own=p.n_own,
                     n_halo=p.n_halo, recv=dict(p.recv), send={q: self._dev(idx) for q, idx in p.send.items()})
            # windows that touch no halo row can be attended while the exchange is in flight
            needs_halo = (p.tok >= p.n_own).any(axis=1)
            for name, sel in (("interior", ~needs_halo), ("boundary", needs_halo)):
                d[name] = None
                if sel.any():
                    d[name] = (self._dev(p.tok[sel]), None if p.grp is None else self._dev(p.grp[sel]))
            self._plan_cache[key] = d
        return self._plan_cache[key]

    def _backbone(self, x_f, x_b, B, patch_res, rollout_step: int, rows=None):
        """Generator: yields `Exchange` requests when sharded (`rows[stage][rank] = (h0, h1)`)."""
        cfg = self.cfg
        bf = self.bb_dtype == BF16
        T_ = self.bb_dtype
        n_enc, n_dec = len(cfg.encoder_depths), len(cfg.decoder_depths)
        all_res, pads = geometry.stage_resolutions(patch_res, n_enc)
        attn_w = self._attn_weights(rollout_step)
        dims = cfg.stage_dims()

        rank = self.shard.rank if rows is not None else 0

        def local_res(stage):
            """(C, owned rows, W) of this rank at a stage (the whole grid when un-sharded)."""
            C, H, W = all_res[stage]
            if rows is None:
                return (C, H, W)
            h0, h1 = rows[stage][rank]
            return (C, h1 - h0, W)

        def run_blocks(blocks, x_f, x_b, stage, final_out=None):
            res = all_res[stage]
            C, H, W = local_res(stage)
            Ls = C * H * W
            M = B * Ls
            for bi, blk in enumerate(blocks):
                dim, heads = blk["dim"], blk["heads"]
                a_in = x_b if bf else x_f
                w_qkv, w_proj = attn_w[blk["prefix"]]
                if rows is None:
                    qkv = lib.linear(a_in, w_qkv, blk["qkv.b"], self.empty(M, 3 * dim, dtype=T_))
                    tok, grp = self._tables(res, blk["shifted"])
                    ao = lib.window_attention(qkv, blk["qkv.b"], self.empty(M, dim, dtype=T_), tok, grp, B, Ls, dim,
                                              heads)
                else:
                    pl = self._plans(res, blk["shifted"], rows[stage])
                    assert pl["n_own"] == Ls
                    qkv = self.empty(Ls + pl["n_halo"], 3 * dim, dtype=T_)
                    lib.linear(a_in, w_qkv, blk["qkv.b"], qkv[:Ls])
                    ao = self.empty(M, dim, dtype=T_)
                    if pl["send"] or pl["recv"]:
                        # Halo rows travel while the windows that need none of them are attended.  A halo row is only
                        # ever a key / value (its own rank computes its queries), so the k | v columns travel, not q:
                        # two thirds of the bytes.
                        sends = [(q, lib.gather_rows(qkv[:Ls, dim:], idx, self.empty(idx.numel(), 2 * dim, dtype=T_)))
                                 for q, idx in pl["send"].items()]
                        landing = {q: self.empty(cnt, 2 * dim, dtype=T_) for q, (off, cnt) in pl["recv"].items()}
                        yield Exchange(sends, list(landing.items()), deferred=True)
                        if pl["interior"] is not None:
                            lib.window_attention(qkv, blk["qkv.b"], ao, *pl["interior"], B, Ls + pl["n_halo"], dim, heads,
                                                 L_out=Ls)
                        yield Complete()
                        for q, (off, cnt) in pl["recv"].items():
                            lib.copy2d(landing[q], qkv[Ls + off:Ls + off + cnt, dim:])
                        if pl["boundary"] is not None:
                            lib.window_attention(qkv, blk["qkv.b"], ao, *pl["boundary"], B, Ls + pl["n_halo"], dim, heads,
                                                 L_out=Ls)
                    else:
                        lib.window_attention(qkv, blk["qkv.b"], ao, pl["tok"], pl["grp"], B, Ls + pl["n_halo"], dim,
                                             heads, L_out=Ls)
                del qkv
                # D = 512 under autocast: linear + AdaLN + residual in one launch (as the C-ABI handle sequences it)
                # -- when the 128-row tiles fill their rounds of one tile per CU (a latitude band's 270 tiles on 256 CUs
                # would take two rounds for the work of 1.05)
                cus = torch.cuda.get_device_properties(self.device).multi_processor_count
                fuse_env = os.environ.get("AURORA_FUSE_LN", "1")   # 0 never, 1 by the fill rule, 2 always (tests)
                fuse = bf and dim == 512 and (fuse_env == "2" or (fuse_env == "1" and fused_ln_fills(M, cus)))
                if fuse:
                    lib.linear_layernorm(ao, w_proj, blk["proj.b"], blk["norm1.gain"], blk["norm1.shift"], x_f, x_f, x_b)
                    del ao
                else:
                    y = lib.linear(ao, w_proj, blk["proj.b"], self.empty(M, dim, dtype=T_))
                    del ao
                    lib.layernorm(y, blk["norm1.gain"], blk["norm1.shift"], res=x_f, out_f32=x_f, out_t=x_b)
                    del y
                hid = lib.linear(a_in, blk["fc1.w"], blk["fc1.b"], self.empty(M, blk["fc1.w"].shape[0], dtype=T_),
                                 act=lib.ACT_GELU)
                last = final_out is not None and bi == len(blocks) - 1
                if fuse:
                    lib.linear_layernorm(hid, blk["fc2.w"], blk["fc2.b"], blk["norm2.gain"], blk["norm2.shift"], x_f,
                                         final_out if last else x_f, None if last else x_b)
                    del hid
                else:
                    y = lib.linear(hid, blk["fc2.w"], blk["fc2.b"], self.empty(M, dim, dtype=T_))
                    del hid
                    lib.layernorm(y, blk["norm2.gain"], blk["norm2.shift"], res=x_f,
                                  out_f32=final_out if last else x_f, out_t=None if last else x_b)
                    del y
            return x_f, x_b

        by_layer = lambda part, i: [b for b in self.blocks if b["part"] == part and b["layer"] == i]  # noqa: E731

        skips = []
        for i in range(n_enc):
            x_f, x_b = yield from run_blocks(by_layer("enc", i), x_f, x_b, i)
            skips.append(x_f)
            if self.debug_hook:
                self.debug_hook(f"enc{i}", x_f)
            if i < n_enc - 1:
                assert all_res[i][1] > 1 and all_res[i][2] > 1, f"grid {all_res[i]} too small to merge"
                C, H, W = local_res(i)
                m = self.merges[i]
                H2, W2 = (H + 1) // 2, (W + 1) // 2
                M2 = B * C * H2 * W2
                mg = lib.merge_ln(x_f, m["ln_w"], m["ln_b"], self.empty(M2, 4 * dims[i], dtype=T_), B, C, H, W, dims[i])
                nf = self.empty(M2, dims[i + 1])
                if bf:
                    nb = self.empty(M2, dims[i + 1], dtype=BF16)
                    lib.linear(mg, m["w"], None, nb, out2=nf)
                else:
                    nb = None
                    lib.linear(mg, m["w"], None, nf)
                del mg
                x_f, x_b = nf, nb
                if self.debug_hook:
                    self.debug_hook(f"merge{i}", x_f)

        D0 = dims[0]
        L0 = int(np.prod(local_res(0)))
        x_cat = self.empty(B * L0, 2 * D0)
        for i in range(n_dec):
            idx = n_dec - 1 - i
            last_layer = i == n_dec - 1
            final_out = x_cat[:, :D0] if last_layer else None
            blocks = by_layer("dec", i)
            x_f, x_b = yield from run_blocks(blocks, x_f, x_b, idx, final_out=final_out)
            if last_layer and not blocks:
                lib.copy2d(x_f, x_cat[:, :D0])
            if self.debug_hook:
                self.debug_hook(f"dec{i}", x_cat[:, :D0] if last_layer else x_f)
            if i < n_dec - 1:
                C, H, W = local_res(idx)
                s = self.splits[i]
                dim = dims[idx]
                a_in = x_b if bf else x_f
                y1 = lib.linear(a_in, s["w1"], None, self.empty(B * C * H * W, 2 * dim, dtype=T_))
                crop = pads[idx - 1]
                if rows is not None and rank != len(rows[idx]) - 1:
                    crop = (crop[0], 0, crop[2])  # the odd bottom row belongs to the last band only
                Ho, Wo = 2 * H - crop[1], 2 * W - crop[2]
                assert (C, Ho, Wo) == tuple(local_res(idx - 1))
                M2 = B * C * Ho * Wo
                sp = lib.split_ln(y1, s["ln_w"], s["ln_b"], self.empty(M2, dim // 2, dtype=T_), B, C, H, W,
                                  dim // 2, crop[1], crop[2])
                del y1
                # additive skip after the intermediate decoder stages (swin3d.py:930-932): the
                # reference adds skips[index-1] after decoder layer i for 0 < i < n_dec-1, i.e. to
                # the up-sampled output of layer i.  Layer i's up-sampling is this GEMM.
                add_skip = 0 < i < n_dec - 1
                res = skips[idx - 1] if add_skip else None
                nf = self.empty(M2, dim // 2)
                if bf:
                    nb = self.empty(M2, dim // 2, dtype=BF16)
                    lib.linear(sp, s["w2"], None, nb, out2=nf, residual=res)
                else:
                    nb = None
                    lib.linear(sp, s["w2"], None, nf, residual=res)
                del sp
                x_f, x_b = nf, nb
                if self.debug_hook:
                    self.debug_hook(f"split{i}", x_f)
        lib.copy2d(skips[0], x_cat[:, D0:])
        return x_cat

    # -- decoder ------------------------------------------------------------------------------
    @staticmethod
    def _dest(out, kind: int, name: str, shape, device) -> torch.Tensor:
        """The caller's destination tensor for a predicted variable if it is usable as is, else a fresh one."""
        t = None if out is None else out[kind].get(name)
        if (t is not None and tuple(t.shape) == tuple(shape) and t.dtype == F32 and t.is_contiguous()
                and t.device == device):
            return t
        return torch.empty(shape, dtype=F32, device=device)

    def _decode(self, x_cat, batch: Batch, B, H, W, Hp, Wp, levels, out=None) -> Batch:
        cfg, model = self.cfg, self.model
        P, D2 = cfg.patch_size, 2 * cfg.embed_dim
        L, Cl, CA = Hp * Wp, cfg.latent_levels, len(levels)
        md = batch.metadata
        lv = self._levels(levels)
        new_step = md.rollout_step + 1
        clamp_now = new_step >= 1 if cfg.clamp_at_first_step else new_step > 1
        diff = type(model)._predict_difference_history_dim_lookup if model.variant == "air_pollution" else {}

        surf_in, atmos_in = tuple(batch.surf_vars), tuple(batch.atmos_vars)
        wave = model.variant == "wave"
        if wave:
            chans = self._wave_channels(surf_in)
            surf_heads = tuple(n for n, _, _ in chans)
            # output order of the reference's post hook: kept variables, then the directions
            kept = tuple(n for n, s_, _ in chans if n == s_)
            surf_in = kept + tuple(a for a in model.angle_surf_vars if f"{a}_sin" in surf_heads and a not in kept)
        else:
            surf_heads = surf_in + tuple(f"{n}_mod" for n in surf_in if n in cfg.modulation_heads)
        atmos_heads = atmos_in + tuple(f"{n}_mod" for n in atmos_in if n in cfg.modulation_heads)
        P2 = P * P

        # ---- surface heads on latent level 0 ----
        w_sh, b_sh = self._head_weights("surf", surf_heads, levels)
        n_s = len(surf_heads) * P2
        ld_s = _round_up(n_s, 4)
        y_s = self.empty(B * L, ld_s)
        for b in range(B):
            lib.linear(x_cat[b * Cl * L:b * Cl * L + L], w_sh, b_sh, y_s[b * L:(b + 1) * L], n=n_s)
        out_s = [self._dest(out, 0, n, (B, 1, H, W), self.device) for n in surf_in]
        descs = []
        for i, n in enumerate(surf_in):
            loc, sc, _ = self._stats("surf", n, levels)
            first = f"{n}_sin" if (wave and n not in surf_heads) else n
            d = lib.unpatch_var(out_s[i].data_ptr(), loc.data_ptr(), sc.data_ptr(),
                                int(clamp_now and n in cfg.positive_surf_vars), surf_heads.index(first) * P2)
            self._diff_fields(d, n, diff, surf_heads, P2, 0, batch.surf_vars, levels, False)
            if wave:
                if first != n:
                    d.angle_col0 = surf_heads.index(f"{n}_cos") * P2
                if f"{n}_density" in surf_heads:
                    wmb = batch.static_vars["wmb"]
                    assert wmb.stride(-1) == 1
                    d.dens_col0 = surf_heads.index(f"{n}_density") * P2
                    d.mask, d.mask_sh = wmb.data_ptr(), wmb.stride(0)
                    d.mask_thresh = normalisation.surf_affine("wmb", model.surf_stats)[0]  # normalised > 0
            descs.append(d)
        for i in range(0, len(descs), 32):
            lib.unpatchify(y_s, descs[i:i + 32], B, 1, Hp, Wp, P)

        # ---- level de-aggregation ----
        ctx = self.empty(B * (Cl - 1) * L, D2) if B > 1 else None
        if B == 1:
            ctx = x_cat[L:Cl * L]
        else:
            for b in range(B):
                lib.copy2d(x_cat[b * Cl * L + L:(b + 1) * Cl * L], ctx[b * (Cl - 1) * L:(b + 1) * (Cl - 1) * L])
        sep = cfg.dec_separate_perceiver
        groups = {"main": [n for n in atmos_heads if n not in sep]}
        if sep:
            groups["alt"] = [n for n in atmos_heads if n in sep]
        out_a = [self._dest(out, 1, n, (B, 1, CA, H, W), self.device) for n in atmos_in]
        for gname, names in groups.items():
            if not names:
                continue
            lat = self._resampler(self.dec_layers[gname], ctx, q0=lv[f"dec_q.{gname}"], latents0=lv["dec_queries"],
                                  B=B, cols=L, kv_bstride=(Cl - 1) * L, kv_lstride=L, Lq=CA, Lk=Cl - 1,
                                  heads=cfg.num_heads, eps=cfg.perceiver_ln_eps)
            w_ah, b_ah = self._head_weights("atmos", tuple(names), levels)
            lvl_stride = len(names) * P2 if cfg.level_condition else 0
            n_a = w_ah.shape[0]
            y_a = lib.linear(lat, w_ah, b_ah, self.empty(B * L * CA, _round_up(n_a, 4)), n=n_a)
            del lat
            descs = []
            for n in names:
                if n.endswith("_mod") and n[:-4] in diff:
                    continue  # consumed by its base variable
                i = atmos_in.index(n)
                loc, sc, _ = self._stats("atmos", n, levels)
                d = lib.unpatch_var(out_a[i].data_ptr(), loc.data_ptr(), sc.data_ptr(),
                                    int(clamp_now and n in cfg.positive_atmos_vars), names.index(n) * P2)
                d.lvl_stride = lvl_stride
                self._diff_fields(d, n, diff, names, P2, lvl_stride, batch.atmos_vars, levels, True)
                if model.variant == "air_pollution" and cfg.use_lora and n == "so2":
                    d.clamp_max1_levels = sum(1 << c for c, l_ in enumerate(levels) if l_ >= 850)
                descs.append(d)
            for i in range(0, len(descs), 32):
                lib.unpatchify(y_a, descs[i:i + 32], B, CA, Hp, Wp, P)
            self._keep_y = y_a

        surf_out = {n: out_s[i] for i, n in enumerate(surf_in)}               # (B, 1, H, W)
        atmos_out = {n: out_a[i] for i, n in enumerate(atmos_in)}             # (B, 1, C, H, W)
        new_md = derive_metadata(md, lat=md.lat.to(F32), lon=md.lon.to(F32),
                                 time=tuple(t + cfg.timestep for t in md.time), rollout_step=new_step)
        if self._cur_band is not None:
            return BandBatch(surf_out, dict(batch.static_vars), atmos_out, new_md,
                             full_patch_rows=self._cur_band[0], band=self._cur_band[1], rank=self.shard.rank,
                             world=self.shard.world)
        return Batch(surf_out, dict(batch.static_vars), atmos_out, new_md)

    def _diff_fields(self, d, name, diff, head_names, P2, lvl_stride, prev_vars, levels, is_atmos):
        """Air-pollution difference prediction (aurora.py:761-779): fields of the descriptor."""
        if name in diff and f"{name}_mod" in head_names:
            prev = prev_vars[name]
            idx = diff[name]
            d.mod_col0 = head_names.index(f"{name}_mod") * P2
            pv = prev[:, idx]
            d.prev = pv.data_ptr()
            if is_atmos:
                d.prev_sb, d.prev_sc, d.prev_sh = pv.stride(0), pv.stride(1), pv.stride(2)
            else:
                d.prev_sb, d.prev_sc, d.prev_sh = pv.stride(0), 0, pv.stride(1)
            assert pv.stride(-1) == 1
            _, _, inv = self._stats("atmos" if is_atmos else "surf", name, levels)
            d.inv_scale = inv.data_ptr()
        else:
            d.mod_col0 = -1

    def _head_weights(self, kind: str, names: tuple, levels: tuple):
        key = ("head", kind, names, levels if self.cfg.level_condition and kind == "atmos" else None)
        if key not in self._embed_w_cache:
            sd = self._sd
            if kind == "atmos" and self.cfg.level_condition:
                ws = [sd[f"decoder.atmos_heads.{n}.layers.{level_to_str(lv)}.weight"] for lv in levels for n in names]
                bs = [sd[f"decoder.atmos_heads.{n}.layers.{level_to_str(lv)}.bias"] for lv in levels for n in names]
            else:
                ws = [sd[f"decoder.{kind}_heads.{n}.weight"] for n in names]
                bs = [sd[f"decoder.{kind}_heads.{n}.bias"] for n in names]
            self._embed_w_cache[key] = (torch.cat(ws, dim=0).contiguous(), torch.cat(bs, dim=0).contiguous())
        return self._embed_w_cache[key]


class GraphedStep:
    """One forecast step captured as a hipGraph (BASELINE config 3: roll-out with a captured step).

    The ~750 kernel launches of a step are recorded once (`torch.cuda.CUDAGraph`; every kernel of
    libaurora_hip runs on torch's current stream, which is the capture stream) and replayed with one
    host call.  The graph reads its inputs from private static buffers and, as its last nodes,
    shifts the history in place (oldest state out, prediction in), so consecutive `advance()` calls
    ARE the roll-out.  Only the clock-dependent inputs change from step to step; they are written
    into the engine's persistent time buffers before each replay.  The LoRA weight set and the
    positive-variable clamp are baked in: the roll-out re-captures when `Engine.step_signature`
    changes (lora.py:105-129: after `lora_steps`, or after the first step in "from_second" mode;
    aurora.py:368-388: clamping starts at the second step).
    """

    def __init__(self, engine: Engine, batch: Batch) -> None:
        assert engine.shard is None or engine.shard.world == 1, "graph capture of sharded steps is not supported"
        self.engine = engine
        cfg = engine.cfg
        batch = engine.model.batch_transform_hook(batch)
        batch = batch.type(F32).crop(cfg.patch_size).to(engine.device)
        clone = lambda d_: {k: v.clone() for k, v in d_.items()}  # noqa: E731
        self.state = Batch(clone(batch.surf_vars), clone(batch.static_vars), clone(batch.atmos_vars),
                           derive_metadata(batch.metadata, lat=batch.metadata.lat.clone(),
                                           lon=batch.metadata.lon.clone()))
        self.signature = engine.step_signature(self.state.metadata.rollout_step)
        B = next(iter(self.state.surf_vars.values())).shape[0]
        # Warm every cache (tables, grids, weight sets, allocator pools) with an eager step, then capture.
        engine.step(self.state)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        engine._time_inputs(self.state.metadata.time, B)
        engine._capturing = True
        try:
            with torch.cuda.graph(self.graph):
                self.pred = engine.step(self.state)
                # History shift, inside the graph: the captured step reads fixed addresses, so the states move, not the
                # window (slot t <- slot t+1 in ascending order: no temporary; then the prediction into the last slot).
                for preds, states in ((self.pred.surf_vars, self.state.surf_vars),
                                      (self.pred.atmos_vars, self.state.atmos_vars)):
                    for k, v in preds.items():
                        x = states[k]
                        for t in range(x.shape[1] - 1):
                            x[:, t].copy_(x[:, t + 1])
                        x[:, -1:].copy_(v)
        finally:
            engine._capturing = False

    def advance(self) -> Batch:
        """Replay the graph once: returns the prediction (fresh tensors) and moves the state forward."""
        eng, md = self.engine, self.state.metadata
        assert eng.step_signature(md.rollout_step) == self.signature, "roll-out phase changed: capture a new graph"
        B = next(iter(self.state.surf_vars.values())).shape[0]
        eng._time_inputs(md.time, B)          # the only step-dependent inputs
        self.graph.replay()
        new_md = derive_metadata(self.pred.metadata, time=tuple(t + eng.cfg.timestep for t in md.time),
                                 rollout_step=md.rollout_step + 1)
        out = Batch({k: v.clone() for k, v in self.pred.surf_vars.items()}, dict(self.pred.static_vars),
                    {k: v.clone() for k, v in self.pred.atmos_vars.items()}, new_md)
        self.state = dataclasses.replace(self.state, metadata=derive_metadata(
            md, time=new_md.time, rollout_step=new_md.rollout_step))
        return out
