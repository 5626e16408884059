"""A training driver of the reference's SHAPE, written here (not copied): it drives this repo's `modules` package through exactly
the calls, in exactly the order and with exactly the keyword arguments, that the reference's train.py makes (train.py:35-322) --
so the drop-in claim can be executed where the reference checkout is absent (the driver's round-end GPU box), and checked for
structure where it is present (tests/test_reference_shape.py parses both files and compares the boundary-call sequences of the
training iteration, the optimizer construction and the evaluation loop).

What is the same as train.py:  seeds; `ti.init(arch=ti.cuda[, half2_vectorization=True])`; `exp_step_factor = 1/256 if scale > 0.5
else 0`; warm-up 256 / update every 16; `MODEL_DICT['ngp'](scale=, pos_encoder_type=, max_res=1024 if scale == 0.5 else 4096,
half_opt=)`; `mark_invisible_cells(K, poses, img_wh)`; `GradScaler(2**16 if half_opt else 2**19)`; apex FusedAdam(lr, eps=1e-15)
with the torch.optim.Adam fall-back; `CosineAnnealingLR(optimizer, max_steps, lr/30)`; per step, inside autocast(fp16):
`update_density_grid(0.01 * MAX_SAMPLES / 3**0.5, warmup=step < warmup_steps)`, `get_rays(direction, pose)`,
`render(model, rays_o, rays_d, exp_step_factor=...)`, `F.mse_loss`, `+ distortion_loss_w * distortion_loss(results).mean()`; then
`zero_grad / scale(loss).backward() / step / update / scheduler.step()`; `torch.save(model.state_dict())`; the evaluation loop with
`render(..., test_time=True, exp_step_factor=...)`.
What is different:  the data.  There is no Synthetic-NeRF Lego here, so the dataset object below renders the analytic Lego-shape
scene of ngp_hip/synthetic.py in memory and hands out batches the way datasets/base.py:34-61 does (`all_images` strategy); PSNR is
computed directly (torchmetrics is not installed); no GUI.

    python examples/train_reference_shape.py --max_steps 300 --wh 200 --n_train 12 --n_test 2 --out run.json
"""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "taichi-nerfs_amd", "compat"), os.path.join(ROOT, "taichi-nerfs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import taichi as ti  # noqa: E402   (compat shim: ti.init / ti.reset are accepted and ignored -- there is no Taichi runtime here)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from modules.distortion import distortion_loss  # noqa: E402
from modules.networks import MODEL_DICT  # noqa: E402
from modules.rendering import MAX_SAMPLES, render  # noqa: E402
from ngp_hip.rays import get_rays  # noqa: E402   (same signature as datasets/ray_utils.py:51-80)


class ProceduralDataset:
    """In-memory stand-in for datasets/nsvf.py on the analytic scene: `.rays [n, H*W, 3]`, `.poses [n, 3, 4]`, `.directions
    [H*W, 3]`, `.K`, `.img_wh`, and `__getitem__` as datasets/base.py:34-61 defines it."""

    def __init__(self, split, n_views, wh, device, seed):
        from ngp_hip.synthetic import procedural_render_gt
        g = torch.Generator().manual_seed(seed)
        z = 0.1 + 0.8 * torch.rand(n_views, generator=g)
        phi = 2 * math.pi * torch.rand(n_views, generator=g)
        r = (1 - z * z).sqrt()
        pos = 1.39 * torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
        fwd = F.normalize(-pos, dim=-1)
        right = F.normalize(torch.cross(fwd, torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd), dim=-1), dim=-1)
        down = torch.cross(fwd, right, dim=-1)
        self.split = split
        self.poses = torch.cat([torch.stack([right, down, fwd], -1), pos[..., None]], -1).to(device)
        focal = 1111.1 * wh / 800
        self.K = torch.tensor([[focal, 0, wh / 2], [0, focal, wh / 2], [0, 0, 1]], device=device)
        self.img_wh = (wh, wh)
        ys, xs = torch.meshgrid(torch.arange(wh, device=device), torch.arange(wh, device=device), indexing="ij")
        self.directions = torch.stack([(xs - wh / 2 + 0.5) / focal, (ys - wh / 2 + 0.5) / focal,
                                       torch.ones_like(xs, dtype=torch.float32)], -1).reshape(-1, 3)
        self.rays = torch.stack([procedural_render_gt(p[:, 3].expand_as(self.directions), self.directions @ p[:, :3].T) for p in self.poses])
        self.batch_size = 8192
        self.ray_sampling_strategy = "all_images"

    def __len__(self):
        return len(self.poses)

    def __getitem__(self, idx):
        if self.split.startswith("train"):
            if self.ray_sampling_strategy == "all_images":
                img_idxs = torch.randint(0, len(self.poses), size=(self.batch_size,), device=self.rays.device)
            else:
                img_idxs = [idx]
            pix_idxs = torch.randint(0, self.img_wh[0] * self.img_wh[1], size=(self.batch_size,), device=self.rays.device)
            rays = self.rays[img_idxs, pix_idxs]
            return {"img_idxs": img_idxs, "pix_idxs": pix_idxs, "pose": self.poses[img_idxs], "direction": self.directions[pix_idxs],
                    "rgb": rays[:, :3]}
        return {"pose": self.poses[idx], "img_idxs": idx, "rgb": self.rays[idx][:, :3]}


def get_opts(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--half_opt", action="store_true", default=False)
    ap.add_argument("--encoder_type", type=str, default="hash")
    ap.add_argument("--distortion_loss_w", type=float, default=0)
    ap.add_argument("--batch_size", type=int, default=8192)
    ap.add_argument("--max_steps", type=int, default=20000)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--wh", type=int, default=200)
    ap.add_argument("--n_train", type=int, default=100)
    ap.add_argument("--n_test", type=int, default=8)
    ap.add_argument("--val_dir", type=str, default="results/")
    ap.add_argument("--out", type=str, default=None)
    return ap.parse_args(argv)


def taichi_init(args):
    ti.init(**dict({"arch": ti.cuda}, **({"half2_vectorization": True} if args.half_opt else {})))


def main(argv=None):
    device = torch.device("cuda")
    for seed_fn in (random.seed, np.random.seed, torch.manual_seed):
        seed_fn(23)
    hparams = get_opts(argv)
    taichi_init(hparams)
    val_dir = hparams.val_dir
    exp_step_factor = 1 / 256 if hparams.scale > 0.5 else 0.
    warmup_steps = 256
    update_interval = 16

    train_dataset = ProceduralDataset("train", hparams.n_train, hparams.wh, device, seed=23)
    train_dataset.batch_size = hparams.batch_size
    train_dataset.ray_sampling_strategy = "all_images"
    test_dataset = ProceduralDataset("test", hparams.n_test, hparams.wh, device, seed=24)

    model_config = dict(scale=hparams.scale, pos_encoder_type=hparams.encoder_type, max_res=1024 if hparams.scale == 0.5 else 4096,
                        half_opt=hparams.half_opt)
    model = MODEL_DICT["ngp"](**model_config).to(device)
    model.mark_invisible_cells(train_dataset.K, train_dataset.poses, train_dataset.img_wh)
    scaler = 2**16 if hparams.half_opt else 2**19                       # (train.py:137-141: "use large scaler, the default is 2**16")
    grad_scaler = torch.cuda.amp.GradScaler(scaler)
    try:
        import apex
        optimizer = apex.optimizers.FusedAdam(model.parameters(), lr=hparams.lr, eps=1e-15)
        optimizer_name = "apex.optimizers.FusedAdam (taichi-nerfs_amd/compat)"
    except ImportError:
        optimizer = torch.optim.Adam(model.parameters(), hparams.lr, eps=1e-15)
        optimizer_name = "torch.optim.Adam"
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, hparams.max_steps, hparams.lr / 30)

    log = []
    torch.cuda.synchronize()
    tic = time.time()
    for step in range(hparams.max_steps + 1):
        model.train()
        i = torch.randint(0, len(train_dataset), (1,)).item()
        data = train_dataset[i]
        direction = data["direction"]
        pose = data["pose"]
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            if step % update_interval == 0:
                model.update_density_grid(0.01 * MAX_SAMPLES / 3**0.5, warmup=step < warmup_steps)
            rays_o, rays_d = get_rays(direction, pose)
            results = render(model, rays_o, rays_d, exp_step_factor=exp_step_factor)
            loss = F.mse_loss(results["rgb"], data["rgb"])
            if hparams.distortion_loss_w > 0:
                loss += hparams.distortion_loss_w * distortion_loss(results).mean()
        optimizer.zero_grad()
        grad_scaler.scale(loss).backward()
        grad_scaler.step(optimizer)
        grad_scaler.update()
        scheduler.step()
        if step % 1000 == 0:
            with torch.no_grad():
                mse = F.mse_loss(results["rgb"], data["rgb"])
                psnr = -10.0 * torch.log(mse) / np.log(10.0)
            log.append((time.time() - tic, step, float(psnr), float(loss), len(data["rgb"]),
                        float(results["rm_samples"] / len(data["rgb"])), float(results["vr_samples"] / len(data["rgb"]))))
            print("elapsed_time=%.2fs | step=%d | psnr=%.2f | loss=%.6f | rays=%d | rm_s=%.1f | vr_s=%.1f" % log[-1])
    torch.cuda.synchronize()
    train_seconds = time.time() - tic

    os.makedirs(val_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(val_dir, "model.pth"))
    with torch.no_grad():
        model.eval()
        directions = test_dataset.directions
        test_psnrs = []
        for test_step in range(len(test_dataset)):
            test_data = test_dataset[test_step]
            rgb_gt = test_data["rgb"]
            poses = test_data["pose"]
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                rays_o, rays_d = get_rays(directions, poses)
                results = render(model, rays_o, rays_d, test_time=True, exp_step_factor=exp_step_factor)
            test_psnrs.append(float(-10.0 * torch.log10(F.mse_loss(results["rgb"].float(), rgb_gt))))
        test_psnr_avg = sum(test_psnrs) / len(test_psnrs)
        print(f"evaluation: psnr_avg={test_psnr_avg}")
    out = {"scene": "procedural Lego-shape (NOT Synthetic-NeRF Lego)", "driver": "examples/train_reference_shape.py", "optimizer": optimizer_name,
           "max_steps": hparams.max_steps, "batch_size": hparams.batch_size, "half_opt": hparams.half_opt,
           "distortion_loss_w": hparams.distortion_loss_w, "train_seconds": train_seconds,
           "train_rays_per_sec": (hparams.max_steps + 1) * hparams.batch_size / train_seconds, "test_psnr_avg": test_psnr_avg,
           "log(elapsed_s,step,psnr,loss,rays,rm_s,vr_s)": log, "checkpoint": os.path.join(val_dir, "model.pth")}
    if hparams.out:
        with open(hparams.out, "w") as f:
            json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    main()
