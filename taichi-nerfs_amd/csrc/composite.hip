// composite.hip -- volume-render compositing (train fwd / closed-form bwd, test-time accumulate) for gfx950.
//
// Replaces modules/volume_train.py:6-48 (+ its Taichi-autodiff backward, :160) and
// modules/volume_render_test.py:4-54 of the reference.
//
// The reference walks each ray serially in one thread and carries transmittance through a global T[] array
// (with a cross-ray race on T[s+1]).  Here one 64-lane wave owns one ray: 64 consecutive samples are loaded
// coalesced, transmittance is a wave-level multiplicative scan with a per-ray carry, the per-ray sums are wave
// reductions, and nothing is shared between rays.  The summation order differs from the serial loop, so these
// kernels are tolerance-checked (1e-5 relative) against the oracle, not bit-checked.
#include "ngp_device.h"
#include <hip/hip_fp16.h>

namespace ngp {

template <bool HALF>
__device__ __forceinline__ void load_rgb(const void* rgbs, size_t s, float c[3]) {
    if (HALF) {
        const __half* p = (const __half*)rgbs + 3 * s;
        c[0] = __half2float(p[0]); c[1] = __half2float(p[1]); c[2] = __half2float(p[2]);
    } else {
        const float* p = (const float*)rgbs + 3 * s;
        c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
    }
}

// ---- a-7 forward (volume_train.py:22-48) --------------------------------------------------------------
template <bool HALF>
__global__ void __launch_bounds__(256) composite_fwd_kernel(const float* __restrict__ sigmas, const void* __restrict__ rgbs,
                                                            const float* __restrict__ deltas, const float* __restrict__ ts,
                                                            const int32_t* __restrict__ rays_a, float thr, int n_rays,
                                                            int32_t* __restrict__ total_samples, float* __restrict__ opacity,
                                                            float* __restrict__ depth, float* __restrict__ rgb,
                                                            float* __restrict__ ws, float* __restrict__ rgb_out, float bg) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int lane = lane_id();
    const int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];
    float T = 1.0f;                       // transmittance entering the current chunk (wave-uniform)
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, op = 0.f;
    int cnt = 0;
    for (int base = 0; base < N; base += NGP_WAVE) {
        const int j = base + lane;
        const bool valid = j < N;
        const size_t s = (size_t)start + j;
        if (!(T > thr)) {                 // everything from here on is behind early termination (:38)
            if (valid) ws[s] = 0.0f;
            continue;
        }
        float a = 0.0f, tm = 0.0f, c[3] = {0.f, 0.f, 0.f};
        if (valid) {
            a = 1.0f - expf(-sigmas[s] * deltas[s]);                                 // :39
            tm = ts[s];
            load_rgb<HALF>(rgbs, s, c);
        }
        const float incl = wave_scan_mul(1.0f - a, lane);                           // prod_{i<=lane} (1-a_i)
        float excl = __shfl_up(incl, 1, NGP_WAVE);
        if (lane == 0) excl = 1.0f;
        const float Ts = T * excl;                                                   // T before this sample
        // liveness as a PREFIX by construction: the Kogge-Stone product is not guaranteed monotone to the last ulp, and
        // ngp_live_compact takes the live samples of a ray to be exactly its first vr[r] ones
        const uint64_t deadm = __ballot(valid && !(Ts > thr));
        const bool live = valid && (deadm == 0 || lane < __builtin_ctzll(deadm));
        const float w = live ? a * Ts : 0.0f;                                        // :40
        if (valid) ws[s] = w;                                                        // :46
        r0 += w * c[0]; r1 += w * c[1]; r2 += w * c[2];                              // :41-43 (per-lane partials)
        dep += w * tm; op += w;                                                      // :44-45
        cnt += live ? 1 : 0;                                                         // :48
        T = deadm ? 0.0f : T * __shfl(incl, NGP_WAVE - 1, NGP_WAVE);    // a dead sample in this chunk ends the ray for good
    }
    r0 = wave_sum(r0); r1 = wave_sum(r1); r2 = wave_sum(r2);
    dep = wave_sum(dep); op = wave_sum(op); cnt = wave_sum_i(cnt);
    if (lane == 0) {
        rgb[3 * ray_idx] = r0; rgb[3 * ray_idx + 1] = r1; rgb[3 * ray_idx + 2] = r2;
        depth[ray_idx] = dep; opacity[ray_idx] = op; total_samples[ray_idx] = cnt;
        if (rgb_out) {                     // rendering.py:219-226: rgb + rgb_bg * (1 - opacity), the colour the loss sees
            const float b = bg * (1.0f - op);
            rgb_out[3 * ray_idx] = r0 + b; rgb_out[3 * ray_idx + 1] = r1 + b; rgb_out[3 * ray_idx + 2] = r2 + b;
        }
    }
}

// ---- a-7 backward: closed form ---------------------------------------------------
//   dL/dc_s     = g_rgb * w_s
//   dL/dsigma_s = delta_s * [ g_rgb.(c_s T+ - (R - rbar_s)) + g_dep (t_s T+ - (D - dbar_s)) + g_op (1 - O)
//                             + g_ws[s] T+ - (W - wbar_s) ]
// with T+ = T_s (1 - a_s), bars = inclusive prefix sums, R/D/O the forward outputs, W = sum_s g_ws[s] w_s.
template <bool HALF>
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ g_op, const float* __restrict__ g_dep,
                                                            const float* __restrict__ g_rgb, const float* __restrict__ g_ws,
                                                            const float* __restrict__ sigmas, const void* __restrict__ rgbs,
                                                            const float* __restrict__ deltas, const float* __restrict__ ts,
                                                            const int32_t* __restrict__ rays_a, const float* __restrict__ opacity,
                                                            const float* __restrict__ depth, const float* __restrict__ rgb,
                                                            const float* __restrict__ ws, float thr, int n_rays,
                                                            float* __restrict__ d_sigmas, void* __restrict__ d_rgbs, float bg) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int lane = lane_id();
    const int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];
    if (N == 0) return;
    const float gr0 = g_rgb[3 * ray_idx], gr1 = g_rgb[3 * ray_idx + 1], gr2 = g_rgb[3 * ray_idx + 2];
    // g_rgb is the gradient of the BLENDED colour rgb + bg (1 - opacity) when bg != 0: the blend's share of d opacity is -bg sum_c g_rgb
    const float gd = g_dep ? g_dep[ray_idx] : 0.0f, go = (g_op ? g_op[ray_idx] : 0.0f) + (bg != 0.0f ? -bg * (gr0 + gr1 + gr2) : 0.0f);
    const float R0 = rgb[3 * ray_idx], R1 = rgb[3 * ray_idx + 1], R2 = rgb[3 * ray_idx + 2];
    const float D = depth[ray_idx], O = opacity[ray_idx];
    float W = 0.0f;
    if (g_ws) {
        for (int j = lane; j < N; j += NGP_WAVE) W += g_ws[(size_t)start + j] * ws[(size_t)start + j];
        W = wave_sum(W);
    }
    float T = 1.0f;
    float cr0 = 0.f, cr1 = 0.f, cr2 = 0.f, cd = 0.f, cw = 0.f;     // carries of the inclusive prefix sums
    for (int base = 0; base < N; base += NGP_WAVE) {
        const int j = base + lane;
        const bool valid = j < N;
        const size_t s = (size_t)start + j;
        if (!(T > thr)) {
            if (valid) {
                d_sigmas[s] = 0.0f;
                if (HALF) { __half* p = (__half*)d_rgbs + 3 * s; p[0] = p[1] = p[2] = __float2half(0.0f); }
                else { float* p = (float*)d_rgbs + 3 * s; p[0] = p[1] = p[2] = 0.0f; }
            }
            continue;
        }
        float a = 0.0f, tm = 0.0f, dl = 0.0f, gw = 0.0f, c[3] = {0.f, 0.f, 0.f};
        if (valid) {
            dl = deltas[s];
            a = 1.0f - expf(-sigmas[s] * dl);
            tm = ts[s];
            load_rgb<HALF>(rgbs, s, c);
            if (g_ws) gw = g_ws[s];
        }
        const float incl = wave_scan_mul(1.0f - a, lane);
        float excl = __shfl_up(incl, 1, NGP_WAVE);
        if (lane == 0) excl = 1.0f;
        const float Ts = T * excl;
        // liveness as a PREFIX by construction: the Kogge-Stone product is not guaranteed monotone to the last ulp, and
        // ngp_live_compact takes the live samples of a ray to be exactly its first vr[r] ones
        const uint64_t deadm = __ballot(valid && !(Ts > thr));
        const bool live = valid && (deadm == 0 || lane < __builtin_ctzll(deadm));
        const float w = live ? a * Ts : 0.0f;
        const float Tp = Ts * (1.0f - a);
        const float p0 = cr0 + wave_scan_add(w * c[0], lane);
        const float p1 = cr1 + wave_scan_add(w * c[1], lane);
        const float p2 = cr2 + wave_scan_add(w * c[2], lane);
        const float pd = cd + wave_scan_add(w * tm, lane);
        float pw = 0.0f;
        if (g_ws) pw = cw + wave_scan_add(gw * w, lane);
        if (valid) {
            float ds = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
            if (live) {
                float acc = gr0 * (c[0] * Tp - (R0 - p0)) + gr1 * (c[1] * Tp - (R1 - p1)) + gr2 * (c[2] * Tp - (R2 - p2));
                acc += gd * (tm * Tp - (D - pd));
                acc += go * (1.0f - O);
                acc += gw * Tp - (W - pw);
                ds = dl * acc;
                dc0 = gr0 * w; dc1 = gr1 * w; dc2 = gr2 * w;
            }
            d_sigmas[s] = ds;
            if (HALF) { __half* p = (__half*)d_rgbs + 3 * s; p[0] = __float2half(dc0); p[1] = __float2half(dc1); p[2] = __float2half(dc2); }
            else { float* p = (float*)d_rgbs + 3 * s; p[0] = dc0; p[1] = dc1; p[2] = dc2; }
        }
        cr0 = __shfl(p0, NGP_WAVE - 1, NGP_WAVE); cr1 = __shfl(p1, NGP_WAVE - 1, NGP_WAVE);
        cr2 = __shfl(p2, NGP_WAVE - 1, NGP_WAVE); cd = __shfl(pd, NGP_WAVE - 1, NGP_WAVE);
        if (g_ws) cw = __shfl(pw, NGP_WAVE - 1, NGP_WAVE);
        T = deadm ? 0.0f : T * __shfl(incl, NGP_WAVE - 1, NGP_WAVE);    // a dead sample in this chunk ends the ray for good
    }
}

// ---- trainer fusion: forward + MSE gradient + backward of one ray in one wave -----------------------------------
// rgb_final = rgb + bg (1 - opacity) (rendering.py:219-226), loss = mean((rgb_final - target)^2) (train.py:193).  The
// gradient of a ray's outputs only needs that ray's own radiance, so the three launches (composite fwd, loss, composite
// bwd) collapse into one: pass 1 composites and leaves R/D/O in registers, pass 2 re-walks the ray (its samples are
// still in L2) with the closed-form backward.  The per-ray squared error goes to sq_err[ray] (nullable) for logging.
// LIVE: the block also appends its rays' live samples (the first vr[r] of each ray: exactly the samples with a non-zero
// gradient) to the compacted list the backward kernels run over -- one returning atomic per 16-ray block on live_total, the
// list is then in block-completion order (rays contiguous), which none of its consumers depends on.  live_zero (the OTHER
// step's counter) is cleared for the next step.  This replaces the separate scan + fill launch (ngp_live_compact, ~20 us on
// the step's critical path) between this kernel and the MLP backward.
template <bool HALF, bool LIVE>
__global__ void __launch_bounds__(LIVE ? 1024 : 256) composite_train_fused_kernel(
    const float* __restrict__ sigmas, const void* __restrict__ rgbs, const float* __restrict__ deltas, const float* __restrict__ ts,
    const int32_t* __restrict__ rays_a, const float* __restrict__ target, float bg, const float* __restrict__ loss_scale, float thr,
    int n_rays, int32_t* __restrict__ vr_per_ray, float* __restrict__ opacity, float* __restrict__ depth, float* __restrict__ rgb,
    float* __restrict__ ws, float* __restrict__ d_sigmas, void* __restrict__ d_rgbs, float* __restrict__ sq_err,
    int32_t* __restrict__ live_idx, int32_t* __restrict__ live_total, int32_t* __restrict__ live_zero) {
    const int n_raw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (!LIVE && n_raw >= n_rays) return;
    const bool has_ray = n_raw < n_rays;                       // LIVE: every wave reaches the block barriers at the end
    const int n = has_ray ? n_raw : 0;
    const int lane = lane_id();
    const int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = has_ray ? rays_a[3 * n + 2] : 0;
    // ---- pass 1: forward (same arithmetic as composite_fwd_kernel)
    float T = 1.0f, r0 = 0.f, r1 = 0.f, r2 = 0.f, dep = 0.f, op = 0.f;
    int cnt = 0;
    for (int base = 0; base < N; base += NGP_WAVE) {
        const int j = base + lane;
        const bool valid = j < N;
        const size_t s = (size_t)start + j;
        if (!(T > thr)) { if (valid) ws[s] = 0.0f; continue; }
        float a = 0.0f, tm = 0.0f, c[3] = {0.f, 0.f, 0.f};
        if (valid) { a = 1.0f - expf(-sigmas[s] * deltas[s]); tm = ts[s]; load_rgb<HALF>(rgbs, s, c); }
        const float incl = wave_scan_mul(1.0f - a, lane);
        float excl = __shfl_up(incl, 1, NGP_WAVE);
        if (lane == 0) excl = 1.0f;
        const float Ts = T * excl;
        // liveness as a PREFIX by construction: the Kogge-Stone product is not guaranteed monotone to the last ulp, and
        // ngp_live_compact takes the live samples of a ray to be exactly its first vr[r] ones
        const uint64_t deadm = __ballot(valid && !(Ts > thr));
        const bool live = valid && (deadm == 0 || lane < __builtin_ctzll(deadm));
        const float w = live ? a * Ts : 0.0f;
        if (valid) ws[s] = w;
        r0 += w * c[0]; r1 += w * c[1]; r2 += w * c[2]; dep += w * tm; op += w; cnt += live ? 1 : 0;
        T = deadm ? 0.0f : T * __shfl(incl, NGP_WAVE - 1, NGP_WAVE);    // a dead sample in this chunk ends the ray for good
    }
    const float R0 = wave_sum(r0), R1 = wave_sum(r1), R2 = wave_sum(r2), D = wave_sum(dep), O = wave_sum(op);
    cnt = wave_sum_i(cnt);
    // ---- MSE gradient of this ray (loss-scaled, like GradScaler.scale(loss).backward())
    const float k = 2.0f / (3.0f * (float)n_rays) * loss_scale[0];
    const float b = bg * (1.0f - O);
    const float e0 = (R0 + b) - target[3 * ray_idx], e1 = (R1 + b) - target[3 * ray_idx + 1], e2 = (R2 + b) - target[3 * ray_idx + 2];
    const float gr0 = k * e0, gr1 = k * e1, gr2 = k * e2;
    const float go = -bg * (gr0 + gr1 + gr2);
    if (lane == 0 && has_ray) {
        rgb[3 * ray_idx] = R0; rgb[3 * ray_idx + 1] = R1; rgb[3 * ray_idx + 2] = R2;
        depth[ray_idx] = D; opacity[ray_idx] = O; vr_per_ray[ray_idx] = cnt;
        // per-ray squared error for logging; summing it here would be 8192 same-address atomics (~12 ns each, measured:
        // 100 us) -- the reader reduces this array instead
        if (sq_err) sq_err[ray_idx] = e0 * e0 + e1 * e1 + e2 * e2;
    }
    // ---- pass 2: closed-form backward (composite_bwd_kernel with g_depth = g_ws = 0)
    T = 1.0f;
    float cr0 = 0.f, cr1 = 0.f, cr2 = 0.f;
    for (int base = 0; base < N; base += NGP_WAVE) {
        const int j = base + lane;
        const bool valid = j < N;
        const size_t s = (size_t)start + j;
        if (!(T > thr)) {
            if (valid) {
                d_sigmas[s] = 0.0f;
                if (HALF) { __half* p = (__half*)d_rgbs + 3 * s; p[0] = p[1] = p[2] = __float2half(0.0f); }
                else { float* p = (float*)d_rgbs + 3 * s; p[0] = p[1] = p[2] = 0.0f; }
            }
            continue;
        }
        float a = 0.0f, dl = 0.0f, c[3] = {0.f, 0.f, 0.f};
        if (valid) { dl = deltas[s]; a = 1.0f - expf(-sigmas[s] * dl); load_rgb<HALF>(rgbs, s, c); }
        const float incl = wave_scan_mul(1.0f - a, lane);
        float excl = __shfl_up(incl, 1, NGP_WAVE);
        if (lane == 0) excl = 1.0f;
        const float Ts = T * excl;
        // liveness as a PREFIX by construction: the Kogge-Stone product is not guaranteed monotone to the last ulp, and
        // ngp_live_compact takes the live samples of a ray to be exactly its first vr[r] ones
        const uint64_t deadm = __ballot(valid && !(Ts > thr));
        const bool live = valid && (deadm == 0 || lane < __builtin_ctzll(deadm));
        const float w = live ? a * Ts : 0.0f;
        const float Tp = Ts * (1.0f - a);
        const float p0 = cr0 + wave_scan_add(w * c[0], lane), p1 = cr1 + wave_scan_add(w * c[1], lane),
                    p2 = cr2 + wave_scan_add(w * c[2], lane);
        if (valid) {
            float ds = 0.0f, dc0 = 0.0f, dc1 = 0.0f, dc2 = 0.0f;
            if (live) {
                float acc = gr0 * (c[0] * Tp - (R0 - p0)) + gr1 * (c[1] * Tp - (R1 - p1)) + gr2 * (c[2] * Tp - (R2 - p2));
                acc += go * (1.0f - O);
                ds = dl * acc;
                dc0 = gr0 * w; dc1 = gr1 * w; dc2 = gr2 * w;
            }
            d_sigmas[s] = ds;
            if (HALF) { __half* p = (__half*)d_rgbs + 3 * s; p[0] = __float2half(dc0); p[1] = __float2half(dc1); p[2] = __float2half(dc2); }
            else { float* p = (float*)d_rgbs + 3 * s; p[0] = dc0; p[1] = dc1; p[2] = dc2; }
        }
        cr0 = __shfl(p0, NGP_WAVE - 1, NGP_WAVE); cr1 = __shfl(p1, NGP_WAVE - 1, NGP_WAVE); cr2 = __shfl(p2, NGP_WAVE - 1, NGP_WAVE);
        T = deadm ? 0.0f : T * __shfl(incl, NGP_WAVE - 1, NGP_WAVE);    // a dead sample in this chunk ends the ray for good
    }
    if (LIVE) {
        __shared__ int s_off[16];
        const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        if (lane == 0) s_off[wave] = has_ray ? cnt : 0;
        __syncthreads();
        if (wave == 0) {
            const int c = lane < nw ? s_off[lane] : 0;
            const int inc = wave_scan_add_i(c, lane);
            int base = 0;
            if (lane == NGP_WAVE - 1 && inc > 0) base = atomicAdd(live_total, inc);
            base = __shfl(base, NGP_WAVE - 1, NGP_WAVE);
            if (lane < nw) s_off[lane] = base + inc - c;
        }
        __syncthreads();
        if (has_ray) {
            const int b = s_off[wave];
            for (int k = lane; k < cnt; k += NGP_WAVE) live_idx[b + k] = start + k;
        }
        if (live_zero && blockIdx.x == 0 && threadIdx.x == 0) *live_zero = 0;
    }
}

// ---- a-8 test-time compositing (volume_render_test.py:18-54): one lane per alive ray, serial over <= steps ----
template <bool HALF>
__global__ void __launch_bounds__(256) composite_test_kernel(const float* __restrict__ sigmas, const void* __restrict__ rgbs,
                                                             const float* __restrict__ deltas, const float* __restrict__ ts,
                                                             const int64_t* __restrict__ pack_info, int64_t* __restrict__ alive,
                                                             float thr, int n_alive, float* __restrict__ opacity,
                                                             float* __restrict__ depth, float* __restrict__ rgb) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int64_t start = pack_info[2 * n], steps = pack_info[2 * n + 1], r = alive[n];
    if (steps == 0) { alive[n] = -1; return; }                                       // :22-23
    float T = 1.0f - opacity[r];                                                     // :25
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, op = 0.f;
    for (int64_t j = 0; j < steps; ++j) {
        const size_t s = (size_t)(start + j);
        const float delta = deltas[s];
        const float a = 1.0f - expf(-sigmas[s] * delta);                             // :33-34
        const float w = a * T;                                                       // :36
        float c[3];
        load_rgb<HALF>(rgbs, s, c);
        c0 += w * c[0]; c1 += w * c[1]; c2 += w * c[2];                              // :41
        dep += w * ts[s]; op += w;                                                   // :42-43
        T *= 1.0f - a;                                                               // :44
        if (T <= thr) { alive[n] = -1; break; }                                      // :46-48
    }
    rgb[3 * r] += c0; rgb[3 * r + 1] += c1; rgb[3 * r + 2] += c2;                    // :50-52
    depth[r] += dep; opacity[r] += op;                                               // :53-54
}

// ---- chunked forward (round 5): which samples are worth shading next ---------------------------------------------------------------
// On scenes whose rays saturate long before their marched samples end (C3: 182 marched, 44 composited samples per ray) the
// reference shades everything the march emitted and then ignores what lies behind T <= 1e-4 (volume_train.py:38; its own
// evaluation loop, rendering.py:62-158, already shades in rounds and drops finished rays).  FusedTrainer does the same for training:
// the samples of a ray are shaded in CHUNKS [begin, begin + len) of its marched range, chunk boundaries at multiples of 64 (the
// compositing kernels read a ray 64 samples at a time and never touch a 64-sample group once T <= thr at its start), and a ray
// whose transmittance after the chunks shaded so far is at or below `thr_stop` gets no further chunk.  thr_stop = thr / 2: the
// estimate here is exp(-sum sigma delta), the compositing kernels form the running product of (1 - a) -- equal up to rounding, so
// every group they read has been shaded; what lies behind is never read, never contributes (its gradients are exact zeros in the
// reference too).
// Block shape of the two list builders below: 16 lanes per ray, 64 rays per 1024-thread block, ONE returning atomic per block on the
// list's counter.  (First version: a wave per ray, 16 rays per block = 4096 same-address atomics per launch at 65 536 rays -- ~12 ns
// each: 50 us of a launch that moves 12 MB; profiles/r05_rocprofv3_garden_timed_region.txt.)
constexpr int LIST_LANES = 16, LIST_RPB = 1024 / LIST_LANES;

// appends `c` consecutive sample indices first .. first + c of this 16-lane group's ray to list (block-wise range allocation)
__device__ __forceinline__ void list_append_block(int c, int first, int32_t* __restrict__ list, int32_t* __restrict__ count, int* s_off) {
    const int grp = threadIdx.x / LIST_LANES, sub = threadIdx.x % LIST_LANES, lane = lane_id();
    if (sub == 0) s_off[grp] = c;
    __syncthreads();
    if (threadIdx.x < NGP_WAVE) {                    // wave 0: LIST_RPB == 64 counts, one per lane
        const int v = s_off[lane];
        const int inc = wave_scan_add_i(v, lane);
        int base = 0;
        if (lane == NGP_WAVE - 1 && inc > 0) base = atomicAdd(count, inc);
        base = __shfl(base, NGP_WAVE - 1, NGP_WAVE);
        s_off[lane] = base + inc - v;
    }
    __syncthreads();
    const int b = s_off[grp];
    for (int k = sub; k < c; k += LIST_LANES) list[b + k] = first + k;
}

__global__ void __launch_bounds__(1024) chunk_schedule_kernel(const int32_t* __restrict__ rays_a, const float* __restrict__ sigmas,
                                                              const float* __restrict__ deltas, int n_rays, int begin, int len,
                                                              int prev_begin, float thr_stop, float* __restrict__ T_state,
                                                              int32_t* __restrict__ list, int32_t* __restrict__ count,
                                                              int32_t* __restrict__ count_zero) {
    static_assert(LIST_RPB == NGP_WAVE, "wave 0 scans the block's ray counts with one wave scan");
    __shared__ int s_off[LIST_RPB];
    const int grp = threadIdx.x / LIST_LANES, sub = threadIdx.x % LIST_LANES;
    const int n = blockIdx.x * LIST_RPB + grp;
    if (blockIdx.x == 0 && threadIdx.x == 0 && count_zero) *count_zero = 0;
    int start = 0, c = 0;
    if (n < n_rays) {
        start = rays_a[3 * n + 1];
        const int N = rays_a[3 * n + 2];
        float T = 1.0f;
        if (begin > 0) {
            T = T_state[n];
            if (T > 0.0f) {
                float sum = 0.0f;
                const int hi = min(begin, N);
                for (int j = prev_begin + sub; j < hi; j += LIST_LANES) sum += sigmas[(size_t)start + j] * deltas[(size_t)start + j];
#pragma unroll
                for (int d = 1; d < LIST_LANES; d <<= 1) sum += __shfl_xor(sum, d, NGP_WAVE);      // (stays inside the 16-lane group)
                T = T * expf(-sum);
                if (!(T > thr_stop)) T = 0.0f;                  // (NaN included: the compositing kernels stop at a NaN as well)
            }
        }
        if (sub == 0) T_state[n] = T;
        c = (T > 0.0f && begin < N) ? min(len, N - begin) : 0;
    }
    list_append_block(c, start + begin, list, count, s_off);
}

// The live-sample list of ngp_live_compact in block-completion order (each ray contiguous; none of the *_live kernels depends on the
// order): the first vr_per_ray[r] samples of every ray.  One launch that moves the list once -- ngp_live_compact's ray-ORDERED list
// costs every block a scan over all rays (152 us at 65 536 rays) and stays what the deterministic mode uses.
__global__ void __launch_bounds__(1024) live_list_kernel(const int32_t* __restrict__ rays_a, const int32_t* __restrict__ vr_per_ray,
                                                         int n_rays, int32_t* __restrict__ live_idx, int32_t* __restrict__ live_total,
                                                         int32_t* __restrict__ live_zero) {
    __shared__ int s_off[LIST_RPB];
    const int grp = threadIdx.x / LIST_LANES;
    const int n = blockIdx.x * LIST_RPB + grp;
    if (blockIdx.x == 0 && threadIdx.x == 0 && live_zero) *live_zero = 0;
    int start = 0, c = 0;
    if (n < n_rays) { start = rays_a[3 * n + 1]; c = vr_per_ray[rays_a[3 * n]]; }
    list_append_block(c, start, live_idx, live_total, s_off);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_composite_train_fwd_bg(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                               const int32_t* rays_a, float T_threshold, int n_rays, int32_t* total_samples, float* opacity,
                               float* depth, float* rgb, float* ws, float* rgb_out, float bg, void* stream) {
    if (n_rays <= 0) return 0;
    dim3 grid((n_rays + 3) / 4), block(256);
    if (rgbs_is_half)
        hipLaunchKernelGGL(composite_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts, rays_a,
                           T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, rgb_out, bg);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts, rays_a,
                           T_threshold, n_rays, total_samples, opacity, depth, rgb, ws, rgb_out, bg);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_composite_train_fwd(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                            const int32_t* rays_a, float T_threshold, int n_rays, int32_t* total_samples, float* opacity,
                            float* depth, float* rgb, float* ws, void* stream) {
    return ngp_composite_train_fwd_bg(sigmas, rgbs, rgbs_is_half, deltas, ts, rays_a, T_threshold, n_rays, total_samples, opacity, depth,
                                      rgb, ws, nullptr, 0.0f, stream);
}

int ngp_composite_train_bwd_bg(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                               const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                               const int32_t* rays_a, const float* opacity, const float* depth, const float* rgb, const float* ws,
                               float T_threshold, int n_rays, float* dL_dsigmas, void* dL_drgbs, float bg, void* stream) {
    if (n_rays <= 0) return 0;
    dim3 grid((n_rays + 3) / 4), block(256);
    if (rgbs_is_half)
        hipLaunchKernelGGL(composite_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, dL_dopacity, dL_ddepth, dL_drgb,
                           dL_dws, sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws, T_threshold, n_rays, dL_dsigmas,
                           dL_drgbs, bg);
    else
        hipLaunchKernelGGL(composite_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, dL_dopacity, dL_ddepth, dL_drgb,
                           dL_dws, sigmas, rgbs, deltas, ts, rays_a, opacity, depth, rgb, ws, T_threshold, n_rays, dL_dsigmas,
                           dL_drgbs, bg);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_composite_train_bwd(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                            const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                            const int32_t* rays_a, const float* opacity, const float* depth, const float* rgb, const float* ws,
                            float T_threshold, int n_rays, float* dL_dsigmas, void* dL_drgbs, void* stream) {
    return ngp_composite_train_bwd_bg(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, rgbs_is_half, deltas, ts, rays_a, opacity, depth,
                                      rgb, ws, T_threshold, n_rays, dL_dsigmas, dL_drgbs, 0.0f, stream);
}

int ngp_composite_train_fused_live(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                                   const int32_t* rays_a, const float* target, float bg, const float* loss_scale, float T_threshold,
                                   int n_rays, int32_t* vr_per_ray, float* opacity, float* depth, float* rgb, float* ws,
                                   float* d_sigmas, void* d_rgbs, float* sq_err, int32_t* live_idx, int32_t* live_total,
                                   int32_t* live_zero, void* stream) {
    if (n_rays <= 0) return 0;
    if (!live_idx) {
        dim3 grid((n_rays + 3) / 4), block(256);
        if (rgbs_is_half)
            hipLaunchKernelGGL((composite_train_fused_kernel<true, false>), grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts,
                               rays_a, target, bg, loss_scale, T_threshold, n_rays, vr_per_ray, opacity, depth, rgb, ws, d_sigmas, d_rgbs,
                               sq_err, nullptr, nullptr, nullptr);
        else
            hipLaunchKernelGGL((composite_train_fused_kernel<false, false>), grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts,
                               rays_a, target, bg, loss_scale, T_threshold, n_rays, vr_per_ray, opacity, depth, rgb, ws, d_sigmas, d_rgbs,
                               sq_err, nullptr, nullptr, nullptr);
    } else {
        if (!live_total) return -1;
        dim3 grid((n_rays + 15) / 16), block(1024);
        if (rgbs_is_half)
            hipLaunchKernelGGL((composite_train_fused_kernel<true, true>), grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts,
                               rays_a, target, bg, loss_scale, T_threshold, n_rays, vr_per_ray, opacity, depth, rgb, ws, d_sigmas, d_rgbs,
                               sq_err, live_idx, live_total, live_zero);
        else
            hipLaunchKernelGGL((composite_train_fused_kernel<false, true>), grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts,
                               rays_a, target, bg, loss_scale, T_threshold, n_rays, vr_per_ray, opacity, depth, rgb, ws, d_sigmas, d_rgbs,
                               sq_err, live_idx, live_total, live_zero);
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_composite_train_fused(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                              const int32_t* rays_a, const float* target, float bg, const float* loss_scale, float T_threshold,
                              int n_rays, int32_t* vr_per_ray, float* opacity, float* depth, float* rgb, float* ws,
                              float* d_sigmas, void* d_rgbs, float* sq_err, void* stream) {
    return ngp_composite_train_fused_live(sigmas, rgbs, rgbs_is_half, deltas, ts, rays_a, target, bg, loss_scale, T_threshold, n_rays,
                                          vr_per_ray, opacity, depth, rgb, ws, d_sigmas, d_rgbs, sq_err, nullptr, nullptr, nullptr, stream);
}

int ngp_composite_test(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                       const int64_t* pack_info, int64_t* alive_indices, float T_threshold, int n_alive, float* opacity,
                       float* depth, float* rgb, void* stream) {
    if (n_alive <= 0) return 0;
    dim3 grid((n_alive + 255) / 256), block(256);
    if (rgbs_is_half)
        hipLaunchKernelGGL(composite_test_kernel<true>, grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts, pack_info,
                           alive_indices, T_threshold, n_alive, opacity, depth, rgb);
    else
        hipLaunchKernelGGL(composite_test_kernel<false>, grid, block, 0, (hipStream_t)stream, sigmas, rgbs, deltas, ts, pack_info,
                           alive_indices, T_threshold, n_alive, opacity, depth, rgb);
    NGP_LAUNCH_CHECK();
    return 0;
}

// Round 5, chunked forward: append samples [begin, begin + len) of every ray that is still alive to `list` (count[0] += their
// number; count_zero[0] = 0 for a later launch; both nullable only for count_zero).  begin > 0: the ray's transmittance state
// T_state[row of rays_a] is first advanced over [prev_begin, begin) (which must have been shaded) and the ray retired when it is
// <= thr_stop; begin == 0 initialises the state.  begin, len, prev_begin: multiples of 64.
int ngp_chunk_schedule(const int32_t* rays_a, const float* sigmas, const float* deltas, int n_rays, int begin, int len, int prev_begin,
                       float thr_stop, float* T_state, int32_t* list, int32_t* count, int32_t* count_zero, void* stream) {
    if (n_rays <= 0) return 0;
    if (!rays_a || !T_state || !list || !count || begin < 0 || len <= 0 || prev_begin < 0 || prev_begin > begin) return -1;
    if ((begin | len | prev_begin) & 63) return -1;
    if (begin > 0 && (!sigmas || !deltas)) return -1;
    hipLaunchKernelGGL(chunk_schedule_kernel, dim3((n_rays + LIST_RPB - 1) / LIST_RPB), dim3(1024), 0, (hipStream_t)stream, rays_a, sigmas, deltas, n_rays,
                       begin, len, prev_begin, thr_stop, T_state, list, count, count_zero);
    NGP_LAUNCH_CHECK();
    return 0;
}

// The live-sample list (the first vr_per_ray[r] samples of ray r, rays_a row order irrelevant) in block-completion order:
// live_total[0] must be 0 at launch (live_zero, nullable, is cleared for a caller alternating two counters).  Same content as
// ngp_live_compact's list, other order; one launch, one atomic per 64 rays.
int ngp_live_list(const int32_t* rays_a, const int32_t* vr_per_ray, int n_rays, int32_t* live_idx, int32_t* live_total,
                  int32_t* live_zero, void* stream) {
    if (n_rays <= 0) return 0;
    if (!rays_a || !vr_per_ray || !live_idx || !live_total) return -1;
    hipLaunchKernelGGL(live_list_kernel, dim3((n_rays + LIST_RPB - 1) / LIST_RPB), dim3(1024), 0, (hipStream_t)stream, rays_a, vr_per_ray,
                       n_rays, live_idx, live_total, live_zero);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
