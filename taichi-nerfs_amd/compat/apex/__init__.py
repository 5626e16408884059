"""`import apex` for the reference's unchanged train.py (train.py:143-149 tries `apex.optimizers.FusedAdam` first and falls back
to torch.optim.Adam): the one class it uses, backed by this repo's one-pass Adam kernel.  Not NVIDIA apex."""
from . import optimizers  # noqa: F401
