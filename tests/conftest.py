import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lego_bitfield():
    """Trained-Lego 128^3 occupancy bitfield (3.94 % occupied); data fixture shipped by the reference under
    deployment/InstantNGP/taichi_ngp/compiled/density_bitfield.bin, stored here compressed."""
    return np.load(os.path.join(GOLDEN, "lego_density_bitfield.npz"))["density_bitfield"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import ngp_oracle
    ngp_oracle.build()
    return ngp_oracle


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the product library; GPU tests must run on the real extension."""
    from ngp_hip import lib
    lib.build()
    return lib.load()


def ray_order(rays_a):
    """Index array that brings per-sample arrays laid out as rays_a says (row = (ray, start, count), ranges in ANY order: the
    reference packs with atomic adds, ngp_march_train_fused in block-completion order) into ray order: x[ray_order(rays_a)] is
    what a ray-ordered packing (the oracle's serial loop, the count / scan / write chain) holds."""
    ra = rays_a.detach().cpu().numpy() if hasattr(rays_a, "detach") else np.asarray(rays_a)
    ra = ra[np.argsort(ra[:, 0], kind="stable")]
    parts = [np.arange(s, s + c, dtype=np.int64) for _, s, c in ra]
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)
