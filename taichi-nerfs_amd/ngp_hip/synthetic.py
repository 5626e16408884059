"""Seeded synthetic workloads of the shapes BASELINE.json names (there is no dataset and no network here).

`lego_rays`  : N rays from 800x800 pinhole cameras (f = 1111.1 px, the Synthetic-NeRF intrinsics) placed on a
               sphere of radius ~1.39 around the origin and looking at it -- what datasets/ray_utils.get_rays of the
               reference hands to render() for Lego (un-normalised directions, one origin per ray).
`garden_rays`: N rays from cameras on a ring inside a scale-16 unbounded scene (360_v2 Garden shape).
`random_bitfield` / `ball_slab_bitfield`: seeded occupancy bitfields for the initialisation / Garden regimes.
`garden_field` / `garden_render_gt`: an analytic UNBOUNDED scene of the 360_v2 Garden shape (object on a table in the unit box,
               ground out to 0.85 * scale, boxes at radii 1.5 .. 10) and its renderer (exponentially spaced samples, black
               background) -- the scene-consistent targets of bench.py --scene garden and of the Garden recipe run.
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


# ---- analytic unbounded scene (Garden shape): everything in MODEL coordinates, z up, the cameras ring at radius ~1 ---------
GARDEN_SIGMA = 80.0


def _garden_far_boxes(scale):
    boxes = []
    for k in range(10):
        r = 1.5 * 1.6 ** (k % 5) * (1.0 if k < 5 else 1.27)
        if r > 0.8 * scale:
            continue
        ang = 2 * np.pi * (k / 10.0) + 0.3
        h = 0.18 * r
        col = (0.25 + 0.7 * ((k * 37) % 10) / 10.0, 0.25 + 0.7 * ((k * 53) % 10) / 10.0, 0.25 + 0.7 * ((k * 71) % 10) / 10.0)
        boxes.append(((r * np.cos(ang), r * np.sin(ang), -0.65 + h), (h, h, h), col))
    return boxes


def garden_field(x, scale=16.0):
    """x: [...,3] torch tensor (model coordinates) -> (sigma [...], rgb [...,3]).  A striped ball on a table inside the unit box,
    a ground slab (z in [-0.75, -0.65]) with a polar checker out to 0.85 * scale, ten boxes growing with their distance."""
    import torch
    X, Y, Z = x[..., 0], x[..., 1], x[..., 2]
    sigma = torch.zeros(x.shape[:-1], device=x.device)
    rgb = torch.zeros(x.shape, device=x.device)

    def put(inside, col):
        nonlocal sigma, rgb
        sigma = torch.where(inside, torch.full_like(sigma, GARDEN_SIGMA), sigma)
        rgb = torch.where(inside[..., None], col, rgb)
    rxy = torch.sqrt(X * X + Y * Y).clamp_min(1e-6)
    ground = (Z > -0.75) & (Z < -0.65) & (rxy < 0.85 * scale)
    check = torch.sign(torch.sin(8.0 * torch.atan2(Y, X)) * torch.sin(5.0 * torch.log(rxy + 0.05)))
    g_col = torch.stack([0.35 + 0.15 * check, 0.5 + 0.2 * check, 0.3 + 0.1 * check], -1)
    put(ground, g_col)
    table = (X.abs() < 0.45) & (Y.abs() < 0.45) & ((Z + 0.4).abs() < 0.05)
    put(table, torch.stack([0.55 + 0.1 * torch.sin(30.0 * X), 0.35 + 0.05 * torch.sin(30.0 * X), 0.2 + 0.0 * X], -1))
    for leg in ((0.38, 0.38), (-0.38, 0.38), (0.38, -0.38), (-0.38, -0.38)):
        put(((X - leg[0]).abs() < 0.04) & ((Y - leg[1]).abs() < 0.04) & (Z > -0.65) & (Z < -0.45),
            torch.tensor([0.3, 0.2, 0.1], device=x.device).expand(x.shape))
    ball = (X * X + Y * Y + (Z + 0.05) ** 2) < 0.3 ** 2
    stripe = 0.5 + 0.5 * torch.sin(25.0 * Z + 8.0 * torch.atan2(Y, X))
    put(ball, torch.stack([0.9 * stripe + 0.05, 0.2 + 0.6 * (1 - stripe), 0.15 + 0.0 * X], -1))
    for c, h, col in _garden_far_boxes(scale):
        inside = ((x - torch.tensor(c, device=x.device, dtype=x.dtype)).abs() < torch.tensor(h, device=x.device, dtype=x.dtype)).all(-1)
        shade = 0.7 + 0.3 * torch.sin(6.0 * (X + Y + Z) / h[0])
        put(inside, torch.tensor(col, device=x.device) * shade[..., None])
    return sigma, rgb


def garden_render_gt(rays_o, rays_d, scale=16.0, n_samples=1024, chunk=8192, near=0.02):
    """Radiance of the analytic Garden-shape scene along rays [N,3] whose origins lie inside [-scale, scale]^3: midpoint rule over
    n_samples EXPONENTIALLY spaced points between `near` and the exit of the box (the march's own spacing grows with t, train.py:54),
    black background (rendering.py:219-226 for exp_step_factor > 0) -> [N,3] float32."""
    import torch
    out = []
    with torch.no_grad():
        for i in range(0, rays_o.shape[0], chunk):
            o, d = rays_o[i:i + chunk].float(), rays_d[i:i + chunk].float()
            inv = 1.0 / d
            far = torch.maximum((-scale - o) * inv, (scale - o) * inv).amin(-1).clamp_min(2 * near)
            ratio = torch.log(far / near)
            u = (torch.arange(n_samples, device=o.device) + 0.5) / n_samples
            ts = near * torch.exp(u[None, :] * ratio[:, None])
            dt = ts * (ratio / n_samples)[:, None] * d.norm(dim=-1, keepdim=True)
            sigma, rgb = garden_field(o[:, None] + ts[..., None] * d[:, None], scale)
            alpha = 1 - torch.exp(-sigma * dt)
            T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha[:, :-1]], 1), 1)
            out.append(((alpha * T)[..., None] * rgb).sum(1))
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
