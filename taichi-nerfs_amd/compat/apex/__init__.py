"""`import apex` for the reference's unchanged train.py (train.py:143-149 tries `apex.optimizers.FusedAdam` first and falls back
to torch.optim.Adam): the one class it uses, backed by this repo's one-pass Adam kernel.  Not NVIDIA apex."""
import os as _os

if _os.environ.get("NGP_NO_APEX") == "1":          # A/B runs: train.py:150-156 then falls back to torch.optim.Adam, as without apex
    raise ImportError("compat apex disabled by NGP_NO_APEX=1")
from . import optimizers  # noqa: F401
