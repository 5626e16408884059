"""Minimal imageio stand-in (PIL-backed): imread / imsave / imwrite as the reference's driver uses them."""
import numpy as np
from PIL import Image


def imread(path):
    return np.asarray(Image.open(path))


def imsave(path, arr):
    Image.fromarray(np.asarray(arr)).save(path)


imwrite = imsave
