"""Golden-vector cases shared by tools/make_golden.py (generator, needs the reference) and
the parity tests (consumers, need only the committed .npz files).

Each case names a model class of the public API, constructor keywords (tiny dimensions),
the input grid and the number of roll-out steps.  Weights and inputs are NOT stored: they
are regenerated bit-identically from oracle/detdata.py.  Only the reference's outputs are
stored (tests/golden/<case>.npz, float32 casts of an fp64 reference run).
"""

from datetime import timedelta

_TINY = dict(
    embed_dim=64,
    num_heads=2,
    encoder_depths=(2, 2, 2),
    encoder_num_heads=(1, 2, 4),
    decoder_depths=(2, 2, 2),
    decoder_num_heads=(4, 2, 1),
)

CASES = {
    # Window padding at every stage, odd patch-merge sizes (13x26 -> 7x13 -> 4x7), clamped
    # windows at the last stage, latitude crop 53 -> 52, LoRA "single", two roll-out steps.
    "base_pad": dict(
        cls="Aurora", kwargs=dict(**_TINY, use_lora=True), H=53, W=104, B=1, T=2,
        levels=(100, 250, 500, 850), steps=2,
    ),
    # README-shaped grid (17x32), batch of two, no LoRA.
    "small_b2": dict(
        cls="AuroraSmallPretrained", kwargs=dict(**_TINY), H=17, W=32, B=2, T=2,
        levels=(100, 250, 500, 850), steps=1,
    ),
    # One LoRA per roll-out step, switched off from step `lora_steps` on.
    "lora_all": dict(
        cls="Aurora", kwargs=dict(**_TINY, use_lora=True, lora_mode="all", lora_steps=2),
        H=16, W=32, B=1, T=2, levels=(100, 250, 500, 850), steps=3,
    ),
    # Wave-style flags: LN on q/k of the level aggregation, LoRA from the second step,
    # 12 h step, checkpoint history 3 with 2 history states given.
    "stabilised_12h": dict(
        cls="Aurora",
        kwargs=dict(**_TINY, use_lora=True, lora_mode="from_second", stabilise_level_agg=True,
                    timestep=timedelta(hours=12), max_history_size=3),
        H=32, W=48, B=1, T=2, levels=(50, 500, 1000), steps=2,
    ),
    # High-res style: patch size 10.
    "patch10": dict(
        cls="AuroraHighRes",
        kwargs=dict(**_TINY), H=41, W=80, B=1, T=2, levels=(100, 250, 500, 850), steps=1,
    ),
    # Air-pollution variant: level-conditioned embeddings/heads, dynamic + atmospheric static
    # variables, second decoder Perceiver, modulation heads, pre/post hooks, clamping.
    "air_pollution": dict(
        cls="AuroraAirPollution",
        kwargs=dict(**_TINY), H=25, W=48, B=1, T=2, levels=(50, 500, 850, 1000), steps=2,
    ),
    # Ocean-wave variant: wind speed/direction split, NaN marking of absent wave systems, density
    # channels, sin/cos of directions and their inverse, water-body masking; LoRA from the second step.
    "wave": dict(
        cls="AuroraWave",
        kwargs=dict(**_TINY), H=17, W=32, B=1, T=2, levels=(100, 250, 500, 850), steps=2,
    ),
}
