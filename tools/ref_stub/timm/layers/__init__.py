import torch


def _ntuple(n):
    def f(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x,) * n

    return f


to_2tuple = _ntuple(2)
to_3tuple = _ntuple(3)


class DropPath(torch.nn.Module):
    """Identity at inference (drop_path is 0.0 in every published config)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return x
