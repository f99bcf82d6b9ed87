"""The C ABI from a host program without Python: examples/c_host/aurora_forecast.c (plain C99 + the HIP runtime's C API)
is compiled with gcc against include/aurora_hip.h and the in-tree library, run as its own process on a case directory
(packed weight file, raw float32 inputs, a text description of the Aurora.__init__ keywords and the grid), and its
roll-out must match the reference goldens to the tolerance of the Python-facing API.  The CPU half of this file checks
that the example builds and links against every symbol it uses.
"""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from aurora_amd import normalisation
from tests import helpers

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "examples" / "c_host" / "aurora_forecast.c"
LIBDIR = ROOT / "aurora_amd" / "_lib"
_MODES = {"single": 0, "from_second": 1, "all": 2}


def build_host(out_dir: Path, rccl: bool = False) -> Path:
    """`rccl`: the band mode too -- the halo transport over librccl (examples/c_host/rccl_transport.c)."""
    exe = out_dir / ("aurora_forecast_rccl" if rccl else "aurora_forecast")
    extra_src = [str(SRC.parent / "rccl_transport.c")] if rccl else []
    extra = ["-DAURORA_WITH_RCCL", "-D_POSIX_C_SOURCE=200809L", f"-I{SRC.parent}"] if rccl else []
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
           *extra, str(SRC), *extra_src, "-o", str(exe), f"-L{LIBDIR}", "-laurora_hip", "-L/opt/rocm/lib", "-lamdhip64",
           *(["-lrccl"] if rccl else []), "-lm", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_c_host_example_builds_and_links(tmp_path):
    exe = build_host(tmp_path)
    res = subprocess.run([str(exe)], capture_output=True, text=True)     # no arguments: usage + the library's ABI version
    assert res.returncode == 2 and "usage: aurora_forecast" in res.stderr and "ABI version" in res.stderr


def test_c_host_band_mode_builds_and_links_against_rccl(tmp_path):
    """A latitude band with no Python in the process: the C host with the RCCL transport (ncclSend / ncclRecv in one
    ncclGroup per exchange on a side stream, events to and from the launch stream) compiles warning-free and links against
    librccl and every band entry point of the library it uses.  It cannot RUN here or on the one-GPU box (RCCL refuses two
    ranks on one device): the self-test it performs before its first step is what a multi-GPU host would see first."""
    exe = build_host(tmp_path, rccl=True)
    for args in ([], [str(tmp_path), "--world", "2"], [str(tmp_path), "--rank", "2", "--world", "2", "--nccl-id", "x"]):
        res = subprocess.run([str(exe), *args], capture_output=True, text=True)
        assert res.returncode == 2 and "--nccl-id FILE" in res.stderr, (args, res.stderr)
    plain = build_host(tmp_path)
    res = subprocess.run([str(plain), str(tmp_path), "--rank", "0", "--world", "2", "--nccl-id", "x"], capture_output=True, text=True)
    assert res.returncode == 1 and "without -DAURORA_WITH_RCCL" in res.stderr


def test_rccl_loopback_driver_builds(tmp_path):
    """examples/c_host/rccl_loopback.c (the one-rank loop-back of the C transport that tests/test_gpu_sharded.py runs on the
    GPU box) compiles warning-free and links here too; without a GPU it stops at device selection."""
    exe = tmp_path / "rccl_loopback"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}",
           f"-I{SRC.parent}", "-I/opt/rocm/include", str(SRC.parent / "rccl_loopback.c"), str(SRC.parent / "rccl_transport.c"),
           "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    res = subprocess.run([str(exe), str(tmp_path / "id")], capture_output=True, text=True, timeout=120)
    assert res.returncode in (0, 1, 77), (res.returncode, res.stderr[-500:])   # no device here: hipSetDevice fails (1); 0 on a GPU box


def write_case(path: Path, case, cfg, surf, static, atmos, lat, lon, times) -> None:
    levels = tuple(case["levels"])
    lst = lambda v: f"{len(v)} " + " ".join(str(x) for x in v)  # noqa: E731
    num = lambda v: f"{len(v)} " + " ".join(repr(float(x)) for x in v)  # noqa: E731
    sa = [normalisation.surf_affine(n) for n in cfg.surf_vars]
    ta = [normalisation.surf_affine(n) for n in cfg.static_vars]
    aa = [normalisation.atmos_affine(n, levels) for n in cfg.atmos_vars]
    rec = {
        "embed_dim": cfg.embed_dim, "patch_size": cfg.patch_size, "latent_levels": cfg.latent_levels, "num_heads": cfg.num_heads,
        "encoder_depths": lst(cfg.encoder_depths), "encoder_heads": lst(cfg.encoder_num_heads),
        "decoder_depths": lst(cfg.decoder_depths), "decoder_heads": lst(cfg.decoder_num_heads), "window": lst(cfg.window_size),
        "enc_depth": cfg.enc_depth, "dec_depth": cfg.dec_depth, "perceiver_ln_eps": repr(cfg.perceiver_ln_eps),
        "max_history": cfg.max_history_size, "timestep_hours": repr(cfg.timestep.total_seconds() / 3600),
        "stabilise_level_agg": int(cfg.stabilise_level_agg), "use_lora": int(cfg.use_lora), "lora_steps": cfg.lora_steps,
        "lora_mode": _MODES[cfg.lora_mode], "autocast": 0,
        "surf_vars": lst(cfg.surf_vars), "static_vars": lst(cfg.static_vars), "atmos_vars": lst(cfg.atmos_vars),
        "lat": num(lat.tolist()), "lon": num(lon.tolist()), "levels": num(levels),
        "levels_float32": int(not all(isinstance(v, int) for v in levels)),
        "B": case["B"], "T": case["T"], "steps": case["steps"], "time_hours": num([t.timestamp() / 3600 for t in times]),
        "surf_loc": num([a[0] for a in sa]), "surf_scale": num([a[1] for a in sa]),
        "static_loc": num([a[0] for a in ta]), "static_scale": num([a[1] for a in ta]),
        "atmos_loc": num([x for a in aa for x in a[0]]), "atmos_scale": num([x for a in aa for x in a[1]]),
    }
    (path / "case.txt").write_text("".join(f"{k} {v}\n" for k, v in rec.items()))
    for kind, d in (("surf", surf), ("static", static), ("atmos", atmos)):
        for n, t in d.items():
            np.ascontiguousarray(t.float().numpy(), "<f4").tofile(path / f"{kind}_{n}.f32")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["base_pad", "small_b2", "patch10"])
def test_c_host_rollout_matches_reference_golden(name, tmp_path):
    from tests.test_gpu_handle import Handle

    case, meta = helpers.case_model_meta(name)
    cfg = meta.config
    sd = {k: v.numpy() for k, v in helpers.case_state_dict(meta, torch.float32).items()}
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    write_case(tmp_path, case, cfg, surf, static, atmos, lat, lon, times)
    h = Handle(cfg, autocast=False, state_dict=sd)       # only to write the packed weight file the C host loads
    h.check(h.L.aurora_hip_save_packed(h.h, str(tmp_path / "weights.aurorahip").encode(), None))
    h.close()
    exe = build_host(tmp_path)
    res = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    print(res.stdout.strip(), file=sys.stderr)
    gold = helpers.load_golden(name)
    worst = 0.0
    for step in range(case["steps"]):
        for kind, names in (("surf", cfg.surf_vars), ("atmos", cfg.atmos_vars)):
            for n in names:
                ref = torch.from_numpy(gold[f"s{step}.{kind}.{n}"])
                got = torch.from_numpy(np.fromfile(tmp_path / f"pred{step}_{kind}_{n}.f32", "<f4")).reshape(ref.shape)
                e = helpers.mean_rel_err(got, ref)
                worst = max(worst, e)
                assert e <= 1e-4 and helpers.rel_err(got, ref) <= 1e-3, (step, kind, n, e)
    print(name, "C host roll-out worst mean-rel", worst, file=sys.stderr)
