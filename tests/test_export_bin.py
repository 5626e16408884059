"""`.bin` blobs of the reference's mobile exporter (ngp_hip/export.py; reference taichi_ngp.py:34-84, utils.cpp:100-176):
header layout, every dtype code, the six blobs made from a deployment.npy written by modules.utils.save_deployment_model."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from ngp_hip import export  # noqa: E402


@pytest.mark.parametrize("dt,code", [(np.float32, 0), (np.float16, 1), (np.int32, 2), (np.int16, 3), (np.uint32, 4), (np.uint16, 5)])
def test_blob_header_and_round_trip(tmp_path, dt, code):
    a = (np.arange(37) * 3 - 11).astype(dt).reshape(37, 1)
    path = export.write_bin(tmp_path / "x.bin", a)
    raw = open(path, "rb").read()
    assert np.frombuffer(raw[:8], np.int32).tolist() == [code, 37] and len(raw) == 8 + 37 * np.dtype(dt).itemsize
    b = export.read_bin(path)
    assert b.dtype == np.dtype(dt) and np.array_equal(b, a.reshape(-1))


def test_blob_rejects_other_dtypes_and_bad_files(tmp_path):
    with pytest.raises(TypeError):
        export.write_bin(tmp_path / "d.bin", np.zeros(3, np.float64))
    (tmp_path / "short.bin").write_bytes(b"\x00\x00")
    with pytest.raises(ValueError):
        export.read_bin(tmp_path / "short.bin")
    (tmp_path / "trunc.bin").write_bytes(np.array([0, 5], np.int32).tobytes() + b"\x00" * 8)
    with pytest.raises(ValueError):
        export.read_bin(tmp_path / "trunc.bin")


def test_export_of_a_saved_deployment_model(tmp_path, lego_bitfield):
    """deployment.npy as modules.utils.save_deployment_model lays it out (reference utils.py:230-253) -> the exporter's blobs."""
    rng = np.random.default_rng(0)
    blob = {"poses": rng.standard_normal((25, 3, 4)).astype(np.float32),
            "model.density_bitfield": lego_bitfield,
            "model.hash_encoder.params": rng.standard_normal(4096).astype(np.float32),
            "model.per_level_scale": 0.2772588722239781,
            "model.xyz_encoder.params": rng.standard_normal(32 * 64 + 64 * 16).astype(np.float32),
            "model.rgb_net.params": rng.standard_normal(32 * 64 + 16 * 16).astype(np.float32)}
    np.save(tmp_path / "deployment.npy", blob)
    files = export.export_deployment_bins(tmp_path / "deployment.npy", tmp_path / "compiled", dtype=np.float16)
    assert sorted(files) == ["density_bitfield", "hash_embedding", "pose", "rgb_weights", "sigma_weights"]
    bits = export.read_bin(files["density_bitfield"])
    assert bits.dtype == np.uint32 and bits.size * 4 == lego_bitfield.size
    assert np.array_equal(bits.view(np.uint8), lego_bitfield.reshape(-1))
    assert int(np.unpackbits(bits.view(np.uint8)).sum()) == 82688                       # the fixture's occupied cells
    emb = export.read_bin(files["hash_embedding"])
    assert emb.dtype == np.float16 and np.array_equal(emb, blob["model.hash_encoder.params"].astype(np.float16))
    pose = export.read_bin(files["pose"])
    assert pose.size == 12 and np.array_equal(pose, blob["poses"][20].astype(np.float16).reshape(-1))
    f32 = export.export_deployment_bins(blob, tmp_path / "c32", dtype=np.float32)
    assert np.array_equal(export.read_bin(f32["sigma_weights"]), blob["model.xyz_encoder.params"])


def test_modules_save_deployment_model_feeds_the_exporter(tmp_path):
    """The drop-in save_deployment_model writes what export_deployment_bins reads (CPU tensors only).  Like the reference's, it
    is written for the 16-wide deployment model (the rgb output layer [3, 16] is padded to [16, 16], utils.py:231-233)."""
    from modules import utils as mutils

    class Lin:
        def __init__(self, o, i):
            self.weight = torch.randn(o, i)

    class Net:
        def __init__(self, hidden, out):
            self.hidden_layers, self.output_layer = [hidden], out

    class Enc:
        hash_table = torch.randn(1000)
        log_b = 0.25

    class Model:
        rgb_net = Net(Lin(16, 32), Lin(3, 16))
        xyz_encoder = Net(Lin(16, 32), Lin(16, 16))
        pos_encoder = Enc()
        density_bitfield = torch.arange(64, dtype=torch.uint8)

    class Data:
        poses = torch.randn(21, 3, 4)

    mutils.save_deployment_model(Model(), Data(), tmp_path)
    files = export.export_deployment_bins(tmp_path / "deployment.npy", tmp_path / "out")
    assert export.read_bin(files["rgb_weights"]).size == 16 * 32 + 16 * 16
    assert export.read_bin(files["density_bitfield"]).size == 16
