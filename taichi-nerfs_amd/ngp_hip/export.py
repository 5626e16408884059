"""The `.bin` weight blobs of the reference's mobile exporter (export compatibility).

The reference turns `deployment.npy` (written by `modules.utils.save_deployment_model`, reference utils.py:230-253) into one
blob per array for its C++ inference harness: `save_aot_weights`, deployment/InstantNGP/taichi_ngp/taichi_ngp.py:34-65, read
back by `utils.cpp:100-176`.  Layout of a blob:

    int32 dtype code | int32 number of elements | flat payload
    dtype codes: 0 float32, 1 float16, 2 int32, 3 int16, 4 uint32, 5 uint16

`export_deployment_bins` writes the six blobs `prepare_aot_files` writes (taichi_ngp.py:67-84: hash_embedding, sigma_weights,
rgb_weights, density_bitfield as uint32 words, pose #20 as 3x4, directions when present), so a model trained through this
package feeds the reference's exporter / mobile demo unchanged.  Host-side only (numpy); nothing here touches the GPU.
"""
import os

import numpy as np

_CODES = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int32): 2, np.dtype(np.int16): 3,
          np.dtype(np.uint32): 4, np.dtype(np.uint16): 5}
_DTYPES = {v: k for k, v in _CODES.items()}


def write_bin(path, arr):
    """One blob (taichi_ngp.py:34-65).  Raises TypeError on a dtype the format has no code for (the reference asserts)."""
    arr = np.ascontiguousarray(arr)
    code = _CODES.get(arr.dtype)
    if code is None:
        raise TypeError("no .bin dtype code for %s (float32, float16, int32, int16, uint32, uint16 only)" % arr.dtype)
    with open(path, "wb") as f:
        f.write(np.array([code, arr.size], dtype=np.int32).tobytes())
        f.write(arr.reshape(-1).tobytes())
    return path


def read_bin(path):
    """Inverse of write_bin (what utils.cpp:100-176 does): flat array of the stored dtype."""
    raw = open(path, "rb").read()
    if len(raw) < 8:
        raise ValueError("%s: shorter than the 8-byte header" % path)
    code, n = (int(v) for v in np.frombuffer(raw[:8], dtype=np.int32))
    if code not in _DTYPES or n < 0:
        raise ValueError("%s: bad header (dtype code %d, %d elements)" % (path, code, n))
    dt = _DTYPES[code]
    if len(raw) - 8 != n * dt.itemsize:
        raise ValueError("%s: payload is %d bytes, header says %d x %s" % (path, len(raw) - 8, n, dt))
    return np.frombuffer(raw[8:], dtype=dt).copy()


def export_deployment_bins(deployment, out_dir, dtype=np.float16, pose_index=20):
    """`deployment`: the dict of deployment.npy or a path to it.  dtype: the exporter's `np_type` (float16 in the shipped demo,
    float32 for app_fp32).  Returns {name: path}."""
    if isinstance(deployment, (str, os.PathLike)):
        deployment = np.load(deployment, allow_pickle=True).item()
    os.makedirs(out_dir, exist_ok=True)
    out = {}

    def put(name, arr):
        out[name] = write_bin(os.path.join(str(out_dir), name + ".bin"), arr)

    put("hash_embedding", np.asarray(deployment["model.hash_encoder.params"]).astype(dtype))
    put("sigma_weights", np.asarray(deployment["model.xyz_encoder.params"]).astype(dtype))
    put("rgb_weights", np.asarray(deployment["model.rgb_net.params"]).astype(dtype))
    put("density_bitfield", np.ascontiguousarray(deployment["model.density_bitfield"]).view(np.uint32))
    poses = np.asarray(deployment["poses"])
    put("pose", poses[min(pose_index, len(poses) - 1)].astype(dtype).reshape(3, 4))
    if "model.directions" in deployment:                       # (the retrain instructions add it; save_deployment_model does not)
        put("directions", np.asarray(deployment["model.directions"]))
    return out
