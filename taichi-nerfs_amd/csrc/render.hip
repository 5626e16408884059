// render.hip -- the fused training render of ngp_hip/fused.py as ONE C entry per direction (round 5).
//
// Replaces, for modules.rendering.render (train path, reference rendering.py:161-228 + networks.py:152-166), the ~9 ctypes launches and
// ~90 pointer marshals per step of the Python autograd node: the forward entry issues [coarse occupancy table] -> one-launch march ->
// hash-grid encode -> MFMA weight repack -> fused MLP forward -> compositing forward, the backward entry compositing backward ->
// live-sample list -> fused MLP backward (weight gradients as per-block slabs) -> LDS-sliced scatter-add whose head sums the slabs (float-atomic
// kernels where the level table does not fit it), on one
// stream, through the same extern "C" entry points the operator path uses.  No new arithmetic lives here: only the launch sequence.
// The argument block is a plain C struct (include/ngp_hip.h: ngp_render_args) the caller keeps per sample arena and patches per call.
#include "ngp_device.h"
#include "../../include/ngp_hip.h"

namespace ngp {
// zero two 16-byte-granular buffers in one launch (the table gradient and the 9408 weight gradients)
__global__ void __launch_bounds__(256) clear2_kernel(uint4* __restrict__ a, long na, uint4* __restrict__ b, long nb) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = i0; i < na; i += stride) a[i] = z;
    for (long i = i0; i < nb; i += stride) b[i] = z;
}
}  // namespace ngp

extern "C" {

int ngp_render_train_fwd(const ngp_render_args* a, void* stream) {
    if (!a || a->n_rays <= 0 || !a->levels) return -1;
    int rc = 0;
    if (a->rebuild_coarse && a->coarse)
        if ((rc = ngp_bitfield_coarsen(a->bitfield, a->cascades, a->grid_size, a->coarse, stream)) != 0) return rc;
    if ((rc = ngp_march_train_fused(a->rays_o, a->rays_d, a->hits_t, a->bitfield, a->coarse, a->noise, a->cascades, a->grid_size, a->scale,
                                    a->exp_step_factor, a->max_samples, a->n_rays, a->stage, a->march_ctr, a->rays_a, a->total, a->xyzs,
                                    a->dirs, a->deltas, a->ts, stream)) != 0) return rc;
    const int cap = (int)a->cap;
    if (a->table_kind == 2)
        rc = ngp_hash_fwd_f16_ex(a->xyzs, (const uint16_t*)a->table, a->levels, cap, a->total, 1, a->lo, a->hi, a->enc_pairs, a->enc, stream);
    else if (a->table_kind == 1)
        rc = ngp_hash_fwd_bf16_ex(a->xyzs, (const uint16_t*)a->table, a->levels, cap, a->total, 1, a->lo, a->hi, a->enc_pairs, a->enc, stream);
    else
        rc = ngp_hash_fwd_f32_ex(a->xyzs, (const float*)a->table, a->levels, cap, a->total, 1, a->lo, a->hi, a->enc_pairs, a->enc, stream);
    if (rc != 0) return rc;
    if ((rc = ngp_mlp_pack(a->w[0], a->w[1], a->w[2], a->w[3], a->w[4], a->enc_pairs, a->wpack, stream)) != 0) return rc;
    if ((rc = ngp_mlp_fwd_ex(a->enc, a->dirs, a->wpack, cap, a->total, a->enc_pairs, a->sigmas, a->rgbs, stream)) != 0) return rc;
    // (round 6: rgb_out = rgb + bg (1 - opacity), the blend of rendering.py:219-226, written by the compositing launch itself)
    return ngp_composite_train_fwd_bg(a->sigmas, a->rgbs, 1, a->deltas, a->ts, a->rays_a, a->T_threshold, a->n_rays, a->vr_per_ray, a->opacity,
                                      a->depth, a->rgb, a->ws, a->rgb_out, a->rgb_out ? a->bg : 0.0f, stream);
}

// dW [9408] and dtable are ACCUMULATED into.  Round 5: the caller hands them over cleared (torch.zeros: a cached allocation + one fill
// kernel each; two hipMemsetAsync calls in here cost the host ~15 us each -- measured, profiles/r05_train_py_cprofile.txt).  Round 6,
// clear_grads != 0: ONE fill launch of this library clears both, issued BEHIND the compositing backward -- the autograd node's prelude
// (profiles/r06_modules_path_timeline.txt: 74 us between the loss's backward kernel and the compositing backward, 17 us of them kernels)
// loses its two torch fills and, with bg in the argument block, the two torch kernels of the blend's gradient.
int ngp_render_train_bwd(const ngp_render_args* a, void* stream) {
    if (!a || a->n_rays <= 0 || !a->levels || !a->dW || !a->dtable || !a->live_idx || !a->live_total || !a->live_off) return -1;
    int rc = 0;
    if ((rc = ngp_composite_train_bwd_bg(a->g_opacity, a->g_depth, a->g_rgb, a->g_ws, a->sigmas, a->rgbs, 1, a->deltas, a->ts, a->rays_a,
                                         a->opacity, a->depth, a->rgb, a->ws, a->T_threshold, a->n_rays, a->d_sigmas, a->d_rgbs, a->bg,
                                         stream)) != 0)
        return rc;
    if (a->clear_grads) {
        if (a->dtable_bytes % 16 != 0) return -1;
        hipLaunchKernelGGL(ngp::clear2_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (uint4*)a->dtable, (long)(a->dtable_bytes / 16),
                           (uint4*)a->dW, (long)(ngp::NGP_MLP_NW * 4 / 16));
        NGP_LAUNCH_CHECK();
    }
    // the live-sample list: ngp_live_list (one atomic per 64 rays, block-completion order) when the caller alternates two counters,
    // round 5's ray-ordered ngp_live_compact otherwise -- the *_live kernels do not depend on the order
    if (a->live_zero) rc = ngp_live_list(a->rays_a, a->vr_per_ray, a->n_rays, a->live_idx, a->live_total, a->live_zero, stream);
    else rc = ngp_live_compact(a->rays_a, a->vr_per_ray, a->n_rays, a->live_off, a->live_idx, a->live_total, stream);
    if (rc != 0) return rc;
    const int cap = (int)a->cap;
    // weight gradients: per-block slabs (plain stores) summed by the head of the scatter-add launch, or round 3's float atomics on dW
    int n_parts = 0;
    if (a->dW_parts && !a->force_atomic) {
        n_parts = ngp_mlp_bwd_live_parts(a->enc, a->dirs, a->wpack, a->d_sigmas, a->d_rgbs, cap, a->live_total, a->live_idx, a->enc_pairs,
                                         a->d_enc, a->dW_parts, nullptr, stream);
        if (n_parts <= 0) return n_parts < 0 ? n_parts : -1;
    } else if ((rc = ngp_mlp_bwd_live(a->enc, a->dirs, a->wpack, a->d_sigmas, a->d_rgbs, cap, a->live_total, a->live_idx, a->enc_pairs,
                                      a->d_enc, a->dW, nullptr, stream)) != 0) return rc;
    const bool half = a->table_kind == 2;
    // scatter-add in its LDS-sliced form (half2 encoder: its fp16 arithmetic into an fp16 gradient table, hash_encoder_half.py:300-306);
    // the float-atomic kernels where the level table does not fit it (-2) or the caller forces them
    rc = a->force_atomic ? -2 : ngp_hash_bwd_sliced_prep(a->xyzs, a->levels, cap, a->live_total, a->live_idx, 1, a->lo, a->hi, a->workspace,
                                                         a->workspace_bytes, stream);
    if (rc == 0) {
        if (n_parts > 0)
            return ngp_hash_bwd_sliced_main_slabs(a->d_enc, a->levels, cap, a->live_total, a->enc_pairs, a->dtable, half ? 1 : 0, nullptr,
                                                  a->workspace, a->workspace_bytes, a->dW_parts, n_parts, a->dW, stream);
        return half ? ngp_hash_bwd_sliced_main_f16(a->d_enc, a->levels, cap, a->live_total, a->enc_pairs, (uint16_t*)a->dtable, nullptr,
                                                   a->workspace, a->workspace_bytes, stream)
                    : ngp_hash_bwd_sliced_main(a->d_enc, a->levels, cap, a->live_total, a->enc_pairs, (float*)a->dtable, nullptr,
                                               a->workspace, a->workspace_bytes, stream);
    }
    if (rc != -2) return rc;
    if (n_parts > 0 && (rc = ngp_mlp_dw_reduce(a->dW_parts, n_parts, a->dW, stream)) != 0) return rc;
    return half ? ngp_hash_bwd_f16_live(a->xyzs, a->d_enc, a->levels, cap, a->live_total, a->live_idx, 1, a->lo, a->hi, a->enc_pairs,
                                        (uint16_t*)a->dtable, nullptr, stream)
                : ngp_hash_bwd_f32_live(a->xyzs, a->d_enc, a->levels, cap, a->live_total, a->live_idx, 1, a->lo, a->hi, a->enc_pairs,
                                        (float*)a->dtable, nullptr, stream);
}

}  // extern "C"
