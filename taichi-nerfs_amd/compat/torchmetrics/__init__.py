"""Minimal stand-ins for the two metrics train.py:79-84 instantiates (same call protocol: __call__/compute/reset/to)."""
import torch
import torch.nn.functional as F


class _Metric:
    def __init__(self, data_range=1.0):
        self.data_range = float(data_range)
        self.reset()

    def to(self, device):
        return self

    def reset(self):
        self._sum, self._n = 0.0, 0

    def compute(self):
        return torch.tensor(self._sum / max(self._n, 1))

    def __call__(self, pred, target):
        v = self._value(pred.float(), target.float())
        self._sum += float(v); self._n += 1
        return v


class PeakSignalNoiseRatio(_Metric):
    def _value(self, pred, target):
        mse = F.mse_loss(pred, target)
        return 10.0 * torch.log10(self.data_range**2 / mse)


class StructuralSimilarityIndexMeasure(_Metric):
    """Gaussian-window SSIM (11x11, sigma 1.5), the torchmetrics default."""

    def _value(self, pred, target):
        c1, c2 = (0.01 * self.data_range)**2, (0.03 * self.data_range)**2
        k = torch.arange(11, dtype=pred.dtype, device=pred.device) - 5
        g = torch.exp(-(k**2) / (2 * 1.5**2)); g = (g / g.sum())
        w = (g[:, None] * g[None, :])[None, None].repeat(pred.shape[1], 1, 1, 1)
        f = lambda x: F.conv2d(x, w, groups=x.shape[1])
        mu_p, mu_t = f(pred), f(target)
        s_pp, s_tt, s_pt = f(pred * pred) - mu_p**2, f(target * target) - mu_t**2, f(pred * target) - mu_p * mu_t
        ssim = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p**2 + mu_t**2 + c1) * (s_pp + s_tt + c2))
        return ssim.mean()
