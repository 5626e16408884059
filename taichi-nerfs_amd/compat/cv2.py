"""Minimal cv2 stand-in (PIL-backed) for datasets/color_utils.py:28 (resize) and modules/utils.py:223-228 (colour map)."""
import numpy as np
from PIL import Image

COLORMAP_TURBO = 20
INTER_LINEAR = 1


def resize(img, dsize, interpolation=INTER_LINEAR):
    w, h = int(dsize[0]), int(dsize[1])
    a = np.asarray(img)
    if a.shape[0] == h and a.shape[1] == w:
        return a
    chans = [np.asarray(Image.fromarray(a[..., c].astype(np.float32), mode="F").resize((w, h), Image.BILINEAR))
             for c in range(a.shape[2])] if a.ndim == 3 else [np.asarray(Image.fromarray(a.astype(np.float32), mode="F").resize((w, h), Image.BILINEAR))]
    out = np.stack(chans, -1) if a.ndim == 3 else chans[0]
    return out.astype(a.dtype)


def applyColorMap(gray, colormap=COLORMAP_TURBO):
    g = np.asarray(gray).astype(np.float32) / 255.0
    r = np.clip(1.5 - np.abs(4 * g - 3), 0, 1); gg = np.clip(1.5 - np.abs(4 * g - 2), 0, 1); b = np.clip(1.5 - np.abs(4 * g - 1), 0, 1)
    return (np.stack([b, gg, r], -1) * 255).astype(np.uint8)
