"""Headline benchmark: forecast-steps/sec of one Aurora forward step (+6 h) on the 0.25-degree
ERA5 grid (721 x 1440, 13 pressure levels, 2 history states), bf16 backbone (autocast=True).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one `Aurora.forward` of the 1.3 B-parameter AuroraPretrained configuration
(BASELINE.json configs[1]) on a synthetic Batch that is already resident in HBM: patch embed +
Perceiver encoder, 48 Swin blocks, Perceiver decoder, unpatchify -- nothing skipped.  Weights are
random (`torch.manual_seed(0)`, zero-initialised AdaLN / LoRA tensors re-randomised), inputs are
`randn` in normalised space mapped to physical units (`torch.manual_seed(1)`).

With N > 1 (one process per GPU, RCCL) the default is STRONG scaling of one forecast: the latitude
rows of the token grid are split into N bands (aurora_amd/engine/partition.py), shifted-window
attention exchanges halo rows with the neighbouring ranks by RCCL point-to-point, the state stays
distributed between steps (`gather_output=False`), `value` = steps of that one forecast per second.
`AURORA_BENCH_MODE=replicas` instead lets every rank advance its own forecast (ensemble members, no
collective on the data path, "weak" scaling, `value` = steps of all ranks per second).

One JSON line on stdout (rank 0), with two extra objects:
  roofline      the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / mean launch time measured
                with HIP events on the launch stream during the timed steps, vs the 2.5 PF dense peak;
                `attention` holds the same for the HBM-bound window-attention kernel (bytes / time
                vs 8 TB/s).
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/aurora_oracle.py) timed on this box's host
                cores on the SAME workload: the full 720 x 1440 grid, the same seeded weights and Batch as the GPU step
                (a 1/16 sub-grid is timed first as a fall-back should the full grid not finish in its budget); plus
                the real reference's full-grid timing measured in the build container (profiles/r05_reference_cpu.json).
  parity_full_grid   the outputs of that oracle run against the GPU step's, per variable mean|out-ref| / mean|ref|
                (the metric of the reference's tests/test_model.py:45-61): the fp32 engine must stay below 1e-4, the
                bf16 (autocast) engine below 5e-3 -- the bench exits non-zero otherwise.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from datetime import datetime, timezone
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

LEVELS = (50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000)
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak (6.29 TB/s measured achievable)
FLOP_PER_STEP = 96.8e12
# A (M x K) + W (N x K) read once, C (M x N) written once, bf16, averaged over the 174 plain bf16 linears of a step
# (the 24 stage-0 proj / fc2 launches run fused with their LayerNorm and are reported separately)
ALGO_BYTES_PER_GEMM_LAUNCH = 530.5e6


def attention_moved_fraction(cfg, Hp: int, Wp: int) -> float:
    """(bytes the window-attention launches of a step really move) / (bytes the contract's figure counts).  The contract
     prices a block at 4 * L_pad * D * 2 B with L_pad the window-PADDED token count; padded positions
    are never read or written (their q = k = v is the bias).  0.25 degree: stage 2 is 16,200 tokens padded to 18,432."""
    import math

    n = len(cfg.encoder_depths)
    wc, wh, ww = cfg.window_size
    moved = padded = 0.0
    c, h, w = cfg.latent_levels, Hp, Wp
    for i in range(n):
        blocks = cfg.encoder_depths[i] + cfg.decoder_depths[n - 1 - i]
        d = cfg.embed_dim << i
        pc, ph, pw = (math.ceil(x / min(ws, x)) * min(ws, x) for x, ws in ((c, wc), (h, wh), (w, ww)))
        moved += blocks * c * h * w * d
        padded += blocks * pc * ph * pw * d
        h, w = (h + 1) // 2, (w + 1) // 2
    return moved / padded


def synthetic_batch(cfg, H, W, seed, device, levels=LEVELS):
    from aurora_amd import Batch, Metadata, normalisation as nz

    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    surf = {k: r(1, 2, H, W) * nz.scales[k] + nz.locations[k] for k in cfg.surf_vars}
    static = {k: r(H, W) * nz.scales[k] + nz.locations[k] for k in cfg.static_vars}
    atmos = {}
    for k in cfg.atmos_vars:
        loc = torch.tensor([nz.locations[f"{k}_{lv}"] for lv in levels])[:, None, None]
        sc = torch.tensor([nz.scales[f"{k}_{lv}"] for lv in levels])[:, None, None]
        atmos[k] = r(1, 2, len(levels), H, W) * sc + loc
    for d_, names in ((surf, cfg.positive_surf_vars), (atmos, cfg.positive_atmos_vars)):   # (variant models only)
        for k in names:
            d_[k] = d_[k].abs()
    md = Metadata(lat=torch.linspace(90, -90, H), lon=torch.linspace(0, 360, W + 1)[:-1],
                  time=(datetime(2020, 6, 1, 12, 0, tzinfo=timezone.utc),), atmos_levels=levels)
    return Batch(surf, static, atmos, md).to(device)


_T0 = time.perf_counter()


def log(msg: str) -> None:
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_model(device, cls_name: str = "AuroraPretrained"):
    import aurora_amd

    torch.manual_seed(0)
    # Build on the GPU: initialising 1.3 B parameters with CPU RNG kernels takes minutes.
    with torch.device(device):
        model = getattr(aurora_amd, cls_name)(autocast=True)
        with torch.no_grad():
            for p in model.parameters():
                if not p.any():  # zero-initialised AdaLN modulation: keep the blocks from being no-ops
                    p.normal_(std=0.02)
    return model.eval()


def _tmp_dir() -> Path:
    """Scratch for the weights / outputs handed between the GPU process and the CPU oracle worker."""
    import tempfile

    shm = Path("/dev/shm")
    try:
        if shm.is_dir() and os.access(shm, os.W_OK):
            import shutil

            if shutil.disk_usage(shm).free > 12 << 30:
                return shm
    except OSError:
        pass
    return Path(tempfile.gettempdir())


def cpu_worker(budget_s: float, threads: int, weights: str, out_path: str, H: int, W: int,
               cls_name: str = "AuroraPretrained") -> None:
    """Subprocess body: the CPU oracle (fp32 port of the reference algorithm) on the bench workload -- the weights the
    GPU step used (`weights`: a torch.save'd state_dict) and the same seeded Batch.  A 1/16 sub-grid first (fall-back
    sample), then the full grid, whose outputs go to `out_path` for the parity check.  One JSON object per finished
    sample on stdout; the last one wins."""
    import aurora_amd
    from aurora_amd import normalisation as nz
    from oracle import aurora_oracle as oracle

    torch.set_num_threads(threads)
    with torch.device("meta"):
        meta = getattr(aurora_amd, cls_name)(autocast=True)
    cfg = meta.config
    sd = torch.load(weights, map_location="cpu", mmap=True, weights_only=True)
    t_start = time.perf_counter()
    P = cfg.patch_size
    Hc = H - H % P
    # fall-back sample: a quarter of the rows / columns, an EVEN number of patch rows (with an odd number the middle
    # patch's mean latitude is a rounding residue of 0, outside the position encoding's range; encoder.py / fourier.py:64-72)
    samples = [((Hc // 4) // (2 * P) * 2 * P, (W // 4) // P * P, False), (H, W, True)]
    for (h, w, full) in samples:
        if h < 2 * P or w < P or (not full and (h, w) == (Hc, W)):
            continue
        b = synthetic_batch(cfg, h, w, 1, "cpu")
        t0 = time.perf_counter()
        with torch.inference_mode():
            o_s, o_a, _ = oracle.forward(sd, cfg, b.surf_vars, b.static_vars, b.atmos_vars, b.metadata.lat,
                                         b.metadata.lon, b.metadata.time, LEVELS, 0, nz.locations, nz.scales,
                                         variant=meta.variant)
        dt = time.perf_counter() - t0
        hc = h - h % cfg.patch_size
        frac = (Hc * W) / (hc * w)
        if full:
            torch.save({"surf": {k: v.contiguous() for k, v in o_s.items()},
                        "atmos": {k: v.contiguous() for k, v in o_a.items()}}, out_path)
            sample = (f"CPU oracle (fp32 port of the reference forward, {cls_name}, the GPU step's weights and "
                      f"Batch) on the full {hc}x{w} grid: one forward in {dt:.1f} s")
        else:
            sample = (f"CPU oracle on a {hc}x{w} sub-grid = 1/{frac:g} of the tokens in {dt:.2f} s; value = that rate / "
                      f"{frac:g} (FALL-BACK: the full grid did not finish in its budget)")
        print(json.dumps({"value": 1.0 / (dt * frac), "unit": "forecast-steps/s", "cores": threads, "kind": "port",
                          "sample": sample, "full_grid": bool(full), "seconds": dt}), flush=True)
        if not full and (time.perf_counter() - t_start) + dt * frac * 1.3 > budget_s:
            break   # the full grid would not fit: keep the fall-back sample


def cpu_baseline(model, H: int, W: int, budget_s: float, cls_name: str = "AuroraPretrained") -> tuple[dict, dict | None]:
    """Run `cpu_worker` in a subprocess with a hard timeout (the 256-thread GPU hosts have stalled for minutes inside CPU
    torch ops).  Returns (cpu_baseline object, oracle outputs of the full grid or None)."""
    import subprocess

    threads = min(os.cpu_count() or 1, 64)
    tmp = _tmp_dir()
    wpath, opath = tmp / f"aurora_bench_weights_{os.getpid()}.pt", tmp / f"aurora_bench_oracle_{os.getpid()}.pt"
    outputs = None
    try:
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, wpath)
        cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-worker", "--cpu-budget", str(budget_s - 30),
               "--cpu-threads", str(threads), "--cpu-weights", str(wpath), "--cpu-out", str(opath), "--grid", f"{H}x{W}", "--cpu-model", cls_name]
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s,
                                 env={**os.environ, "OMP_NUM_THREADS": str(threads), "HIP_VISIBLE_DEVICES": ""})
            out, err = res.stdout, res.stderr
        except subprocess.TimeoutExpired as e:
            dec = lambda x: x.decode() if isinstance(x, bytes) else (x or "")  # noqa: E731
            out, err = dec(e.stdout), dec(e.stderr)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if not lines:
            log("CPU worker produced nothing:\n" + err[-2000:])
            return {"value": None, "unit": "forecast-steps/s", "cores": threads, "kind": "port",
                    "sample": f"no CPU sample finished within {budget_s:.0f} s"}, None
        result = json.loads(lines[-1])
        if result.get("full_grid") and opath.exists():
            outputs = torch.load(opath, map_location="cpu", weights_only=True)
        return result, outputs
    finally:
        for f in (wpath, opath):
            try:
                f.unlink()
            except OSError:
                pass


TOL_FP32, TOL_BF16 = 1e-4, 5e-3   # reference tests/test_model.py:52-61 accepts 1e-4 (smooth) ... 5e-3 (winds, humidity)


def mean_rel_err(pred, ref_surf: dict, ref_atmos: dict) -> dict:
    """Per variable mean|out - ref| / mean|ref| (the metric of the reference's tests/test_model.py:45-61), on the device."""
    out = {}
    for d, rd in ((pred.surf_vars, ref_surf), (pred.atmos_vars, ref_atmos)):
        for k, v in d.items():
            r = rd[k].to(v.device).reshape(v.shape).double()
            out[k] = ((v.double() - r).abs().mean() / (r.abs().mean() + 1e-30)).item()
    return out


def full_grid_parity(model, batch, pred_bf16, oracle_out: dict) -> dict:
    """GPU step vs the CPU oracle on the whole grid: the timed bf16 (autocast) engine, and the fp32 engine on the same
    weights (a second handle; the parameters are shared, not copied)."""
    import aurora_amd

    e16 = mean_rel_err(pred_bf16, oracle_out["surf"], oracle_out["atmos"])
    with torch.device("meta"):
        m32 = type(model)(autocast=False)
    m32.load_state_dict(model.state_dict(), assign=True)
    m32 = m32.eval()
    with torch.inference_mode():
        pred32 = m32.forward(batch)
    torch.cuda.synchronize()
    e32 = mean_rel_err(pred32, oracle_out["surf"], oracle_out["atmos"])
    del m32
    H, W = next(iter(pred32.surf_vars.values())).shape[-2:]
    w32, w16 = max(e32.values()), max(e16.values())
    return {"grid": f"{H}x{W}", "metric": "max over variables of mean|out - oracle| / mean|oracle| "
                                          "(reference tests/test_model.py:45-61)",
            "fp32_engine_vs_oracle": w32, "bf16_engine_vs_oracle": w16, "tol_fp32": TOL_FP32, "tol_bf16": TOL_BF16,
            "per_variable_fp32": e32, "per_variable_bf16": e16, "ok": bool(w32 <= TOL_FP32 and w16 <= TOL_BF16)}


def launch_command(args, argv: list[str]) -> list[str]:
    """`python bench.py --gpus N` without a launcher: the torch.distributed.run command that starts the N ranks."""
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), *argv]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=900.0,
                    help="seconds the CPU oracle leg may take (full grid: ~5 min on 64 cores)")
    ap.add_argument("--grid", default="721x1440", help=argparse.SUPPRESS)   # tests run the same script on a small grid
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-weights", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-out", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-model", default="AuroraPretrained", help=argparse.SUPPRESS)
    args = ap.parse_args()
    GH, GW = map(int, args.grid.split("x"))
    if args.cpu_worker:
        cpu_worker(args.cpu_budget, args.cpu_threads, args.cpu_weights, args.cpu_out, GH, GW, args.cpu_model)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher of the N ranks (one process per GPU)
        import subprocess

        cmd = launch_command(args, sys.argv[1:])
        log("launching " + " ".join(cmd))
        sys.exit(subprocess.run(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from datetime import timedelta

        # Test hooks (single-GPU boxes): AURORA_BENCH_SAME_GPU=1 puts every rank on cuda:0 and
        # AURORA_BENCH_BACKEND=gloo replaces RCCL (which needs one GPU per rank) by host-staged gloo.
        if os.environ.get("AURORA_BENCH_SAME_GPU"):
            local_rank = 0
        elif torch.cuda.device_count() < world:
            raise SystemExit(f"--gpus {world} needs {world} visible GPUs, found {torch.cuda.device_count()}")
        backend = os.environ.get("AURORA_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        try:
            dist.init_process_group(backend, timeout=timedelta(seconds=600), **kw)
        except Exception as e:  # noqa: BLE001  -- one JSON line the driver can record instead of N tracebacks
            if rank == 0:
                print(json.dumps({"metric": "forecast-steps/sec (6h step) 0.25deg ERA5 721x1440x13", "value": None,
                                  "n_gpus": world, "error": f"process-group initialisation ({backend}) failed: {e!r}"}), flush=True)
            raise SystemExit(3)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    control = None   # host-side (gloo) group for agreement between ranks that does not depend on RCCL's state (see below)
    if distributed:
        try:
            control = dist.new_group(backend="gloo")
        except Exception as e:  # noqa: BLE001  -- agreement then runs over the main group, as it did before
            log(f"no gloo control group ({e!r}); using the main process group")
        # every rank must use the SAME group: if the control group failed anywhere, nobody uses it (agreed over the main group,
        # which at this point has just been initialised and has carried nothing yet)
        have = [None] * world
        dist.all_gather_object(have, control is not None)
        if not all(have):
            control = None
    bands_error = None

    mode = os.environ.get("AURORA_BENCH_MODE", "bands") if distributed else "single"
    assert mode in ("bands", "replicas", "single"), mode
    log(f"building model (mode {mode})")
    model = build_model(device)  # same seed on every rank: identical weights
    log("model on device; building batch")
    if mode == "bands":
        # one forecast, identical inputs everywhere; each rank keeps its latitude band resident
        model.configure_sharding(rank, world, gather_output=False)
        full = synthetic_batch(model.config, GH, GW, 1, device)
        batch = model.engine().local_band(full.crop(model.patch_size))
        del full
        # before anything is timed: every rank sends a stamped pattern of a real halo message's size to its neighbours
        # through the production transport and checks what arrived (a wrong rank order / fabric / stream order fails here)
        transport = model.engine().native.transport
        test_error = None
        try:
            transport.selftest(4 << 20)
            # test hook: this rank's check "fails" -- AFTER it took part in the exchange, like a real verification failure
            # (a rank that stayed away from the exchange would leave its neighbours waiting for a message, not failing)
            if os.environ.get("AURORA_BENCH_BREAK_SELFTEST") == str(rank):
                raise RuntimeError("self-test broken on purpose (AURORA_BENCH_BREAK_SELFTEST)")
        except Exception as e:  # noqa: BLE001
            test_error = f"halo transport self-test failed on rank {rank}: {e!r}"
        # every rank must take the same branch: agreement runs over a host-side (gloo) group, which works whatever state the
        # failed point-to-point operation left RCCL in
        errors = [None] * world
        dist.all_gather_object(errors, test_error, group=control)
        bands_error = next((e for e in errors if e), None)
        if bands_error is None:
            log("halo transport self-test ok")
        elif os.environ.get("AURORA_BENCH_NO_FALLBACK"):
            if rank == 0:
                print(json.dumps({"metric": "forecast-steps/sec (6h step) 0.25deg ERA5 721x1440x13", "value": None,
                                  "n_gpus": world, "error": bands_error}), flush=True)
            raise SystemExit(4)
        else:
            # The sharded forecast cannot run on this fabric: rather than no record at all, every rank advances its OWN
            # forecast (no data-path collective) and the line says so -- "scaling": "weak", `bands_error` carries the reason.
            log(f"{bands_error} -- falling back to independent replicas")
            mode = "replicas"
            model.configure_sharding(0, 1)
            del batch, transport
            batch = synthetic_batch(model.config, GH, GW, 1 + rank, device)
    else:
        batch = synthetic_batch(model.config, GH, GW, 1 + (rank if mode == "replicas" else 0), device)
    log("batch on device")

    # After a failed halo self-test RCCL's state is unknown: from there on the ranks meet over the host-side group only
    # (barriers, the elapsed-time reduction), so that the replica fall-back leaves its record whatever the fabric does.
    host_sync = bands_error is not None and control is not None

    def barrier():
        if distributed:
            torch.cuda.synchronize()
            dist.barrier(group=control if host_sync else None)
        torch.cuda.synchronize()

    with torch.inference_mode():
        for i in range(args.warmup):
            pred = model.forward(batch)
            torch.cuda.synchronize()
            log(f"warmup step {i} done")
        barrier()
        # (1) the timed region: K un-instrumented steps -> `value`
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pred = model.forward(batch)
        barrier()
        elapsed = time.perf_counter() - t0
        log(f"timed region done: {elapsed / args.steps * 1e3:.1f} ms/step")
        # (2) the same K steps again with HIP events (on the launch stream, inside the C-ABI handle) around every launch
        # of the roofline kernels -> `roofline`.  Kept out of (1): an event pair keeps a launch from overlapping
        # its neighbours, ~250 pairs per step cost about 1 % of it.
        eng = model.engine()
        eng.profile_start({"linear_bf16", "linear_layernorm_bf16", "window_attention_bf16"})
        for _ in range(args.steps):
            model.forward(batch)
        barrier()
        prof = eng.profile_stop()
        eng.profile_start()          # (3) one more step with events around everything, for the per-kernel breakdown
        model.forward(batch)
        breakdown = eng.profile_stop()
        # (4) sharded runs: K more steps per rank with an event pair around every halo `wait` on the launch stream -- how
        # long a rank's stream stood still for messages (what the exchange did NOT hide) -- next to the rank's own step time
        # measured WITHOUT a barrier between ranks; the first scaling record then says whether a rank is slow or waiting
        per_rank = None
        if mode == "bands":
            tr = eng.native.transport
            tr.time_waits = True
            tr.wait_ms()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                model.forward(batch)
            torch.cuda.synchronize()
            mine = [(time.perf_counter() - t1) / args.steps * 1e3, tr.wait_ms() / args.steps, float(tr.exchanges)]
            tr.time_waits = False
            every = [None] * world
            dist.all_gather_object(every, mine)
            per_rank = {"step_ms": [round(v[0], 3) for v in every], "halo_wait_ms": [round(v[1], 3) for v in every],
                        "compute_ms": [round(v[0] - v[1], 3) for v in every],
                        "max_rank_compute_ms": round(max(v[0] - v[1] for v in every), 3)}
    assert torch.isfinite(pred.surf_vars["2t"]).all()

    if distributed:
        on_host = host_sync or dist.get_backend() != "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=control if host_sync else None)
        elapsed = t.item()

    failed = False
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (1 if mode == "bands" else world) * args.steps / elapsed
        zero = {"launches": 0, "ms": 0.0, "work": 0.0}
        g = prof.get("linear_bf16", zero)
        f = prof.get("linear_layernorm_bf16", zero)
        a = prof.get("window_attention_bf16", zero)
        gemm_tf = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else 0.0
        all_tf = (g["work"] + f["work"]) / ((g["ms"] + f["ms"]) * 1e-3) / 1e12 if g["ms"] + f["ms"] else 0.0
        attn_gbs = a["work"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] else 0.0
        # HBM-side bytes per launch of the dominant kernel come from hardware counters, which only rocprofv3 can collect:
        # the tracked PMC summary of THIS command (tools/profile_round.sh -> tools/pmc_rollup.py), not a live value --
        # `traffic_build` names the commit whose library those passes profiled.
        traffic, traffic_build = None, None
        pmc = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"))
        traffic_stale, stale_sources = None, None
        if pmc:
            import hashlib

            pj = json.loads(pmc[-1].read_text())
            traffic, traffic_build = pj.get("linear_bf16_hbm_bytes_per_launch"), pj.get("build_commit")
            # counters are static evidence of the build they profiled: if the GEMM source has changed since (its SHA-256 is
            # recorded by tools/pmc_rollup.py), the figure is withheld instead of going stale silently
            sha = hashlib.sha256((ROOT / "aurora_amd" / "csrc" / "gemm.hip").read_bytes()).hexdigest()
            traffic_stale = pj.get("gemm_source_sha256") != sha
            if traffic_stale:
                traffic = None
            # every kernel source, not only the GEMM's: which of the profiled files differ from the tree (their rows in the
            # committed rNN_kernel_stats.txt / rNN_pmc_summary.json describe an earlier build)
            stale_sources = sorted(n for n, d in (pj.get("source_sha256") or {}).items()
                                   if n != "libaurora_hip.so" and (not (ROOT / "aurora_amd" / "csrc" / n).exists() or
                                   hashlib.sha256((ROOT / "aurora_amd" / "csrc" / n).read_bytes()).hexdigest() != d))
        per = max(args.steps, 1)
        out = {
            "metric": "forecast-steps/sec (6h step) 0.25deg ERA5 721x1440x13",
            "value": value, "unit": "forecast-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if mode == "bands" else "weak" if mode == "replicas" else "n/a", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"AuroraPretrained(autocast=True) 1.3B, 0.25deg ERA5 {GH}x{GW}, 13 levels, "
                                   "T=2, batch 1 per GPU, one forward step (BASELINE.json configs[1])",
                       "parallelism": {"single": "1 GPU",
                                       "bands": f"one forecast over {world} latitude bands, RCCL halo exchange "
                                                "(3 rows per shifted block and side), state kept distributed",
                                       "replicas": f"replica x{world} (independent forecasts, no data-path "
                                                   "collective)"}[mode]},
            "roofline": {
                "kernel": "linear_kernel_256pp (bf16 MFMA GEMM, ping-pong LDS ring): the plain backbone linears of a step; "
                          "`frac_all_matrix_launches` adds the stage-0 proj / fc2 launches that run fused with their "
                          "AdaLN + residual (linear_layernorm_bf16)", "bound": "mfma",
                "achieved": gemm_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": gemm_tf / PEAK_BF16_TFLOPS, "traffic": traffic,
                "traffic_source": f"profiles/{pmc[-1].name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command; "
                                  "2 x FETCH_SIZE per profiles/r02_fetch_calibration.txt)" if pmc else None,
                "traffic_build": traffic_build, "traffic_stale": traffic_stale, "profile_stale_sources": stale_sources,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_GEMM_LAUNCH,
                "launches_per_step": g["launches"] / per, "ms_per_step": g["ms"] / per,
                "frac_all_matrix_launches": all_tf / PEAK_BF16_TFLOPS,
                "all_matrix_launches": {"achieved": all_tf, "launches_per_step": (g["launches"] + f["launches"]) / per,
                                        "ms_per_step": (g["ms"] + f["ms"]) / per},
                "attention": {"kernel": "window_attention_bf16", "bound": "hbm", "achieved": attn_gbs,
                              "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": attn_gbs / PEAK_HBM_GBS,
                              # the same with the bytes really moved (padded window positions are never touched)
                              "frac_moved_bytes": attn_gbs / PEAK_HBM_GBS * attention_moved_fraction(
                                  model.config, (GH - GH % model.patch_size) // model.patch_size, GW // model.patch_size),
                              "launches_per_step": a["launches"] / per, "ms_per_step": a["ms"] / per},
            },
            "step_tflops": FLOP_PER_STEP / (ms_per_step * 1e-3) / 1e12 if (GH, GW) == (721, 1440) else None,
            "kernel_ms_per_step": {k: v["ms"] for k, v in sorted(breakdown.items())},   # (the extra, un-timed step)
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if bands_error is not None:   # asked for a sharded forecast, measured replicas: the record says why
            out["bands_error"] = bands_error
        # the GPU measurement is safe in the log before the (long) CPU leg starts: a driver that gives up on the CPU oracle
        # still finds it on stderr; stdout carries exactly one line, at the end
        log("GPU-only result: " + json.dumps(out))
        if world == 1 and not args.no_cpu_baseline:
            log("running the CPU oracle on the same weights and Batch (full grid)")
            out["cpu_baseline"], oracle_out = cpu_baseline(model, GH, GW, args.cpu_budget)
            if oracle_out is not None:
                log("comparing the GPU step with the oracle")
                out["parity_full_grid"] = full_grid_parity(model, batch, pred, oracle_out)
                failed = not out["parity_full_grid"]["ok"]
            else:
                out["parity_full_grid"] = None
            # The real reference (microsoft/aurora itself) cannot run on this box (no /root/reference here): its
            # timing, and the port's on the same machine and inputs, come from tools/time_reference.py in the build
            # container (tracked file, full 721 x 1440 grid).
            # (the latest record: round 5 re-timed both on an IDLE container -- 206 s, BASELINE.md's 201 s; round 2's 462 s was
            # taken while the container was compiling)
            refs = sorted((ROOT / "profiles").glob("r*_reference_cpu.json"))
            ref = refs[-1] if refs else ROOT / "profiles" / "r02_reference_cpu.json"
            if ref.exists():
                r = json.loads(ref.read_text())
                out["cpu_baseline"]["reference_measured_elsewhere"] = {
                    "value": r["reference_steps_per_s"], "unit": "forecast-steps/s", "cores": r["threads"],
                    "kind": "reference", "port_over_reference_time": r.get("port_over_reference_time"),
                    "sample": f"microsoft/aurora AuroraPretrained fp32, full {r['grid'][0]}x{r['grid'][1]} grid, one "
                              f"forward in {r['reference_s_per_step']:.0f} s on the build container's {r['threads']} cores "
                              f"(tools/time_reference.py, profiles/{ref.name})"}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier(group=control if host_sync else None)
        dist.destroy_process_group()
    if failed:
        log("FULL-GRID PARITY VIOLATED (see parity_full_grid in the JSON line)")
        sys.exit(1)


if __name__ == "__main__":
    main()
