// distortion.hip -- Mip-NeRF-360 distortion loss (forward + hand-written backward) for gfx950.
//
// Replaces modules/distortion.py:15-119 of the reference: four Taichi kernels (serial per-ray prefix sums, elementwise
// loss, serial per-ray reduce, backward).  The reference's own TODO (distortion.py:4-6) asks for a shared-memory scan;
// here one wave owns a ray, 64 samples per trip, and the inclusive scans of w and w*t are wave scans with a carry.
// Summation order differs from the serial loops: tolerance-checked (1e-5) against the oracle / golden vectors.
#include "ngp_device.h"

namespace ngp {

__global__ void __launch_bounds__(256) distortion_fwd_kernel(const float* __restrict__ ws, const float* __restrict__ deltas,
                                                             const float* __restrict__ ts, const int32_t* __restrict__ rays_a,
                                                             int n_rays, float* __restrict__ loss, float* __restrict__ ws_inc,
                                                             float* __restrict__ wts_inc) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int lane = lane_id();
    const int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];
    float cw = 0.f, cwt = 0.f, acc = 0.f;
    for (int base = 0; base < N; base += NGP_WAVE) {
        const int j = base + lane;
        const bool valid = j < N;
        const size_t s = (size_t)start + j;
        const float w = valid ? ws[s] : 0.f, t = valid ? ts[s] : 0.f, d = valid ? deltas[s] : 0.f;
        const float wt = w * t;                                                       // distortion.py:150
        const float wi = cw + wave_scan_add(w, lane), wti = cwt + wave_scan_add(wt, lane);   // inclusive (:37-42)
        const float we = wi - w, wte = wti - wt;                                      // exclusive (:33-35)
        if (valid) {
            ws_inc[s] = wi; wts_inc[s] = wti;
            acc += 2.f * (wti * we - wi * wte) + (1.f / 3.f) * w * w * d;             // :63
        }
        cw = __shfl(wi, NGP_WAVE - 1, NGP_WAVE);
        cwt = __shfl(wti, NGP_WAVE - 1, NGP_WAVE);
    }
    acc = wave_sum(acc);                                                              // :78-84
    if (lane == 0) loss[ray_idx] = acc;
}

__global__ void __launch_bounds__(256) distortion_bwd_kernel(const float* __restrict__ dL_dloss, const float* __restrict__ ws,
                                                             const float* __restrict__ deltas, const float* __restrict__ ts,
                                                             const float* __restrict__ ws_inc, const float* __restrict__ wts_inc,
                                                             const int32_t* __restrict__ rays_a, int n_rays,
                                                             float* __restrict__ dL_dws) {
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int lane = lane_id();
    const int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];
    if (N <= 0) return;
    const float g = dL_dloss[ray_idx];
    const float ws_sum = ws_inc[(size_t)start + N - 1], wts_sum = wts_inc[(size_t)start + N - 1];   // :104-105
    for (int j = lane; j < N; j += NGP_WAVE) {
        const size_t s = (size_t)start + j;
        const float t = ts[s];
        const float selector = (j == 0) ? 0.f : t * ws_inc[s - 1] - wts_inc[s - 1];                  // :112
        float v = g * 2.f * (selector + (wts_sum - wts_inc[s] - t * (ws_sum - ws_inc[s])));          // :114
        v += g * (2.f / 3.f) * ws[s] * deltas[s];                                                    // :116
        dL_dws[s] = v;
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_distortion_fwd(const float* ws, const float* deltas, const float* ts, const int32_t* rays_a, int n_rays, float* loss,
                       float* ws_inc, float* wts_inc, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(distortion_fwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, ws, deltas, ts, rays_a,
                       n_rays, loss, ws_inc, wts_inc);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_distortion_bwd(const float* dL_dloss, const float* ws, const float* deltas, const float* ts, const float* ws_inc,
                       const float* wts_inc, const int32_t* rays_a, int n_rays, float* dL_dws, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(distortion_bwd_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, dL_dloss, ws, deltas, ts,
                       ws_inc, wts_inc, rays_a, n_rays, dL_dws);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
