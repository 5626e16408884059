"""Mirror of reference modules/ray_march.py: raymarching_train (:126-194) and raymarching_test (:270-334).

Differences that are visible to a caller (both documented in DESIGN.md):
  * rays_a comes back in RAY ORDER with start = exclusive prefix sum (the reference's order depends on atomic
    arrival); the per-ray samples are bit-identical.
  * no N*1024 worst-case output allocations: outputs are sized from the scanned total."""
import torch

from ngp_hip import ops as _ops


def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, grid_size, max_samples):
    # noise to perturb the first sample of each ray (reference :138)
    noise = torch.rand_like(rays_o[:, 0])
    return _ops.march_train(rays_o.contiguous(), rays_d.contiguous(), hits_t.contiguous(), density_bitfield, noise,
                            cascades, scale, exp_step_factor, grid_size, max_samples)


def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor, grid_size,
                     max_samples):
    """Marches every alive ray by at most `max_samples` occupied steps; hits_t[:,0] is advanced IN PLACE so the
    next call resumes.  Returns (packed_info[n_alive,2] = (start, count), ray_indices, deltas, ts), packed."""
    if not hits_t.is_contiguous():
        raise ValueError("hits_t must be contiguous: it is updated in place")
    ray_indices, valid_mask, deltas, ts, samples_counter = _ops.march_test(
        rays_o.contiguous(), rays_d.contiguous(), hits_t, alive_indices.contiguous(), density_bitfield, cascades, scale,
        exp_step_factor, grid_size, max_samples)
    valid_mask = valid_mask.bool()
    cumsum = torch.cumsum(samples_counter, 0)
    counts64 = samples_counter.to(torch.int64)
    packed_info = torch.stack([cumsum - counts64, counts64], dim=-1)
    return packed_info, ray_indices[valid_mask], deltas[valid_mask], ts[valid_mask]
