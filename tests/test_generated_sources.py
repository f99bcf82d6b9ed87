"""Generated kernel text stays in step with its generator: `aurora_amd/csrc/gemm_a4_loop.inc` (the K loop of the four-wave
GEMM tile as one asm statement) is committed -- the build does not run `tools/gen_gemm_asm.py` -- so a change to either must
show up here."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _generator():
    spec = importlib.util.spec_from_file_location("gen_gemm_asm", ROOT / "tools" / "gen_gemm_asm.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gemm_a4_loop_is_what_the_generator_writes():
    gen = _generator()
    committed = (ROOT / "aurora_amd" / "csrc" / "gemm_a4_loop.inc").read_text()
    assert committed == gen.build(0)


def test_gemm_a4_loop_text_is_self_consistent():
    """128 MFMAs and 16 LDS-DMA pieces per K unit, every accumulator written exactly once per fragment pair, and the register
    ranges the kernel declares as clobbered cover everything the text names."""
    import re

    text = _generator().build(0)
    body = text[text.index("A4_TEXT_0"):]
    loop = body[body.index("1:"):] if "1:" in body else body
    assert len(re.findall(r"v_mfma_f32_16x16x32_bf16", loop)) % 128 == 0
    regs = {int(m) for m in re.findall(r"\bv(\d+)\b", body)} | {int(a) for m in re.findall(r"\bv\[(\d+):(\d+)\]", body) for a in m}
    assert max(regs) <= 191 and min(r for r in regs if r >= 48) >= 48   # the loop's own registers: v48 .. v191 (A4_CLOBBERS)
    accs = {int(a) for m in re.findall(r"\ba\[(\d+):(\d+)\]", body) for a in m}
    assert min(accs) == 0 and max(accs) == 255
