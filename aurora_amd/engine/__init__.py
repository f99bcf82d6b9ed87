"""The HIP engine behind `Aurora.forward`: geometry, encodings, weight packing and the step."""


def __getattr__(name):
    if name == "Engine":
        from aurora_amd.engine.engine import Engine

        return Engine
    raise AttributeError(name)
