"""Seeded synthetic workloads of the shapes BASELINE.json names (there is no dataset and no network here).

`lego_rays`  : N rays from 800x800 pinhole cameras (f = 1111.1 px, the Synthetic-NeRF intrinsics) placed on a
               sphere of radius ~1.39 around the origin and looking at it -- what datasets/ray_utils.get_rays of the
               reference hands to render() for Lego (un-normalised directions, one origin per ray).
`garden_rays`: N rays from cameras on a ring inside a scale-16 unbounded scene (360_v2 Garden shape).
`random_bitfield` / `ball_slab_bitfield`: seeded occupancy bitfields for the initialisation / Garden regimes.
`procedural_field` / `procedural_render_gt`: an analytic "Lego-shape" scene (base plate, tower, studs inside
               [-0.35, 0.35]^3, white background) and its dense-integration renderer -- the scene-consistent target
               colours bench.py and examples/train_procedural.py train against.
"""
import numpy as np

# ---- analytic scene: axis-aligned boxes (centre, half-size, rgb) ------------------------------------------------------
PROCEDURAL_BOXES = [
    ((0.0, 0.0, -0.22), (0.30, 0.20, 0.05), (0.85, 0.10, 0.10)),
    ((-0.12, 0.0, -0.02), (0.10, 0.10, 0.15), (0.95, 0.80, 0.10)),
    ((0.14, 0.05, -0.07), (0.08, 0.12, 0.10), (0.10, 0.35, 0.85)),
    ((0.0, -0.12, 0.16), (0.22, 0.04, 0.04), (0.15, 0.70, 0.25)),
] + [((-0.2 + 0.1 * i, -0.1 + 0.1 * j, -0.15), (0.025, 0.025, 0.02), (0.85, 0.10, 0.10)) for i in range(5) for j in range(3)]


def procedural_field(x):
    """x: [...,3] torch tensor (world coordinates) -> (sigma [...], rgb [...,3]); density 400 inside the boxes, striped albedo."""
    import torch
    sigma = torch.zeros(x.shape[:-1], device=x.device)
    rgb = torch.ones(x.shape, device=x.device) * 0.5
    shade = 0.75 + 0.25 * torch.sin(40.0 * x.sum(-1, keepdim=True))
    for c, h, col in PROCEDURAL_BOXES:
        inside = ((x - torch.tensor(c, device=x.device)).abs() < torch.tensor(h, device=x.device)).all(-1)
        sigma = torch.where(inside, torch.full_like(sigma, 400.0), sigma)
        rgb = torch.where(inside[..., None], torch.tensor(col, device=x.device) * shade, rgb)
    return sigma, rgb


def procedural_render_gt(rays_o, rays_d, n_samples=768, chunk=16384, scale=0.5):
    """Ground-truth radiance of the analytic scene along rays [N,3] (torch, any device): midpoint-rule integration of
    n_samples points inside the [-scale, scale]^3 box, white background -> [N,3] float32."""
    import torch
    out = []
    with torch.no_grad():
        for i in range(0, rays_o.shape[0], chunk):
            o, d = rays_o[i:i + chunk].float(), rays_d[i:i + chunk].float()
            inv = 1.0 / d
            t0, t1 = (-scale - o) * inv, (scale - o) * inv
            near = torch.minimum(t0, t1).amax(-1).clamp_min(0.01)
            far = torch.maximum(t0, t1).amin(-1)
            hit = far > near
            span = (far - near).clamp_min(0)
            ts = near[:, None] + span[:, None] * (torch.arange(n_samples, device=o.device) + 0.5) / n_samples
            dt = (span / n_samples)[:, None] * d.norm(dim=-1, keepdim=True)
            sigma, rgb = procedural_field(o[:, None] + ts[..., None] * d[:, None])
            alpha = 1 - torch.exp(-sigma * dt)
            T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha[:, :-1]], 1), 1)
            w = alpha * T * hit[:, None]
            out.append((w[..., None] * rgb).sum(1) + (1 - w.sum(1, keepdim=True)))       # white background
    return torch.cat(out)


def _look_at(cam_pos, target, roll):
    """camera-to-world rotations (OpenCV convention: +z forward, +x right, +y down), with in-plane roll."""
    fwd = target - cam_pos
    fwd /= np.linalg.norm(fwd, axis=-1, keepdims=True)
    up = np.zeros_like(fwd)
    up[:, 2] = 1.0
    right = np.cross(fwd, up)
    bad = np.linalg.norm(right, axis=-1) < 1e-6
    right[bad] = np.array([1.0, 0.0, 0.0])
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    down = np.cross(fwd, right)
    c, s = np.cos(roll)[:, None], np.sin(roll)[:, None]
    right2 = c * right + s * down
    down2 = -s * right + c * down
    return np.stack([right2, down2, fwd], axis=-1)       # [N,3,3], columns = camera axes in world


def lego_rays(n_rays, seed=23, radius=1.39, img_wh=(800, 800), focal=1111.1, n_cams=100):
    rng = np.random.default_rng(seed)
    # cameras on the upper hemisphere like the Blender scenes
    u = rng.random(n_cams)
    phi = rng.random(n_cams) * 2 * np.pi
    z = 0.05 + 0.9 * u
    rxy = np.sqrt(1 - z * z)
    cams = radius * np.stack([rxy * np.cos(phi), rxy * np.sin(phi), z], -1)
    rot = _look_at(cams.copy(), np.zeros_like(cams), rng.random(n_cams) * 0.2 - 0.1)
    cam_id = rng.integers(0, n_cams, n_rays)
    px = rng.random(n_rays) * img_wh[0]
    py = rng.random(n_rays) * img_wh[1]
    d_cam = np.stack([(px - img_wh[0] / 2) / focal, (py - img_wh[1] / 2) / focal, np.ones(n_rays)], -1)
    rays_d = np.einsum('nij,nj->ni', rot[cam_id], d_cam)
    rays_o = cams[cam_id]
    return rays_o.astype(np.float32), rays_d.astype(np.float32)


def garden_rays(n_rays, seed=23, ring_radius=1.0, n_cams=64, img_wh=(1297, 840), focal=960.0):
    rng = np.random.default_rng(seed)
    phi = rng.random(n_cams) * 2 * np.pi
    cams = np.stack([ring_radius * np.cos(phi), ring_radius * np.sin(phi), 0.3 + 0.2 * rng.random(n_cams)], -1)
    tgt = rng.normal(size=(n_cams, 3)) * 0.2
    rot = _look_at(cams.copy(), tgt, rng.random(n_cams) * 0.1 - 0.05)
    cam_id = rng.integers(0, n_cams, n_rays)
    px = rng.random(n_rays) * img_wh[0]
    py = rng.random(n_rays) * img_wh[1]
    d_cam = np.stack([(px - img_wh[0] / 2) / focal, (py - img_wh[1] / 2) / focal, np.ones(n_rays)], -1)
    rays_d = np.einsum('nij,nj->ni', rot[cam_id], d_cam)
    return cams[cam_id].astype(np.float32), rays_d.astype(np.float32)


def random_bitfield(cascades, grid_size=128, fraction=0.5, seed=23):
    rng = np.random.default_rng(seed)
    bits = rng.random(cascades * grid_size**3) < fraction
    return np.packbits(bits, bitorder='little')


def _morton_decode(idx):
    def compact(x):
        x = x & 0x49249249
        x = (x | (x >> 2)) & 0xc30c30c3
        x = (x | (x >> 4)) & 0x0f00f00f
        x = (x | (x >> 8)) & 0xff0000ff
        x = (x | (x >> 16)) & 0x0000ffff
        return x
    idx = idx.astype(np.uint32)
    return compact(idx), compact(idx >> 1), compact(idx >> 2)


def ball_slab_bitfield(cascades, scale, grid_size=128, seed=23, ball_r=0.4, slab_half=0.05, far_fraction=0.005):
    """Multi-cascade occupancy for the Garden-shape config: solid ball + ground slab + sparse far cells."""
    rng = np.random.default_rng(seed)
    G = grid_size
    idx = np.arange(G**3)
    x, y, z = _morton_decode(idx)
    out = []
    for c in range(cascades):
        s = min(2.0**(c - 1), scale)
        cx = ((x + 0.5) / G * 2 - 1) * s
        cy = ((y + 0.5) / G * 2 - 1) * s
        cz = ((z + 0.5) / G * 2 - 1) * s
        occ = (cx * cx + cy * cy + cz * cz < ball_r**2) | ((np.abs(cz) < max(slab_half, s / G)) & (cx * cx + cy * cy < 64.0))
        occ |= rng.random(G**3) < far_fraction
        out.append(occ)
    return np.packbits(np.concatenate(out), bitorder='little')
