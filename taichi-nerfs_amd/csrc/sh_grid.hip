// sh_grid.hip -- spherical-harmonics direction encoding and the occupancy-grid integer utilities for gfx950.
//
// Replaces modules/spherical_harmonics.py:7-42 (+ autodiff backward :92) and modules/utils.py:120-169
// (morton3D, morton3D_invert, packbits) of the reference.
#include "ngp_device.h"

namespace ngp {

// ---- a-6 SH16: one lane per direction, four 16-byte stores per lane --------------------------------------
__global__ void __launch_bounds__(256) sh16_fwd_kernel(const float* __restrict__ dirs, int n, float4* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = dirs[3 * (size_t)i], y = dirs[3 * (size_t)i + 1], z = dirs[3 * (size_t)i + 2];
        const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        float4 a, b, c, d;
        a.x = 0.28209479177387814f;                                   // spherical_harmonics.py:27-42, literal forms
        a.y = -0.48860251190291987f * y;
        a.z = 0.48860251190291987f * z;
        a.w = -0.48860251190291987f * x;
        b.x = 1.0925484305920792f * xy;
        b.y = -1.0925484305920792f * yz;
        b.z = 0.94617469575755997f * z2 - 0.31539156525251999f;
        b.w = -1.0925484305920792f * xz;
        c.x = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        c.y = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        c.z = 2.8906114426405538f * xy * z;
        c.w = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        d.x = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        d.y = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        d.z = 1.4453057213202769f * z * (x2 - y2);
        d.w = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
        float4* o = out + 4 * (size_t)i;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d;
    }
}

// analytic Jacobian^T * dout
__global__ void __launch_bounds__(256) sh16_bwd_kernel(const float* __restrict__ dirs, const float4* __restrict__ dout, int n,
                                                       float* __restrict__ ddirs) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = dirs[3 * (size_t)i], y = dirs[3 * (size_t)i + 1], z = dirs[3 * (size_t)i + 2];
        const float4 ga = dout[4 * (size_t)i], gb = dout[4 * (size_t)i + 1], gc = dout[4 * (size_t)i + 2], gd = dout[4 * (size_t)i + 3];
        const float c1 = 0.48860251190291987f, c2 = 1.0925484305920792f, c6 = 0.94617469575755997f,
                    c8 = 0.54627421529603959f, c9 = 0.59004358992664352f, c10 = 2.8906114426405538f,
                    c11 = 0.45704579946446572f, c12 = 0.3731763325901154f, c14 = 1.4453057213202769f;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        dy += -c1 * ga.y; dz += c1 * ga.z; dx += -c1 * ga.w;
        dx += c2 * y * gb.x; dy += c2 * x * gb.x;
        dy += -c2 * z * gb.y; dz += -c2 * y * gb.y;
        dz += 2.f * c6 * z * gb.z;
        dx += -c2 * z * gb.w; dz += -c2 * x * gb.w;
        dx += 2.f * c8 * x * gc.x; dy += -2.f * c8 * y * gc.x;
        dx += c9 * y * (-6.f * x) * gc.y; dy += c9 * (-3.f * x * x + 3.f * y * y) * gc.y;
        dx += c10 * y * z * gc.z; dy += c10 * x * z * gc.z; dz += c10 * x * y * gc.z;
        dy += c11 * (1.f - 5.f * z * z) * gc.w; dz += c11 * y * (-10.f * z) * gc.w;
        dz += c12 * (15.f * z * z - 3.f) * gd.x;
        dx += c11 * (1.f - 5.f * z * z) * gd.y; dz += c11 * x * (-10.f * z) * gd.y;
        dx += c14 * z * 2.f * x * gd.z; dy += -c14 * z * 2.f * y * gd.z; dz += c14 * (x * x - y * y) * gd.z;
        dx += c9 * (-3.f * x * x + 3.f * y * y) * gd.w; dy += c9 * x * 6.f * y * gd.w;
        ddirs[3 * (size_t)i] = dx; ddirs[3 * (size_t)i + 1] = dy; ddirs[3 * (size_t)i + 2] = dz;
    }
}

// ---- a-10 grid utilities ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) morton3d_kernel(const int32_t* __restrict__ coords, int m, int32_t* __restrict__ indices) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
        indices[i] = (int32_t)morton3d((uint32_t)coords[3 * (size_t)i], (uint32_t)coords[3 * (size_t)i + 1],
                                       (uint32_t)coords[3 * (size_t)i + 2]);                 // utils.py:140-145
}
__global__ void __launch_bounds__(256) morton3d_invert_kernel(const int32_t* __restrict__ indices, int m, int32_t* __restrict__ coords) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t ind = (uint32_t)indices[i];                                           // utils.py:120-127
        coords[3 * (size_t)i] = morton3d_invert1(ind >> 0);
        coords[3 * (size_t)i + 1] = morton3d_invert1(ind >> 1);
        coords[3 * (size_t)i + 2] = morton3d_invert1(ind >> 2);
    }
}
// one lane per output byte: two float4 loads -> 8 compares (utils.py:157-169)
__global__ void __launch_bounds__(256) packbits_kernel(const float4* __restrict__ grid, float thr, int n_bytes, uint8_t* __restrict__ out) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_bytes; n += gridDim.x * blockDim.x) {
        const float4 a = grid[2 * (size_t)n], b = grid[2 * (size_t)n + 1];
        uint32_t bits = 0;
        bits |= (a.x > thr) ? 1u : 0u; bits |= (a.y > thr) ? 2u : 0u; bits |= (a.z > thr) ? 4u : 0u; bits |= (a.w > thr) ? 8u : 0u;
        bits |= (b.x > thr) ? 16u : 0u; bits |= (b.y > thr) ? 32u : 0u; bits |= (b.z > thr) ? 64u : 0u; bits |= (b.w > thr) ? 128u : 0u;
        out[n] = (uint8_t)bits;
    }
}

inline int grid1d(long long work) {
    long long b = (work + 255) / 256;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_sh16_fwd(const float* dirs, int n, float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sh16_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dirs, n, (float4*)out);
    NGP_LAUNCH_CHECK();
    return 0;
}
int ngp_sh16_bwd(const float* dirs, const float* dout, int n, float* ddirs, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sh16_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dirs, (const float4*)dout, n, ddirs);
    NGP_LAUNCH_CHECK();
    return 0;
}
int ngp_morton3d(const int32_t* coords, int m, int32_t* indices, void* stream) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(morton3d_kernel, dim3(grid1d(m)), dim3(256), 0, (hipStream_t)stream, coords, m, indices);
    NGP_LAUNCH_CHECK();
    return 0;
}
int ngp_morton3d_invert(const int32_t* indices, int m, int32_t* coords, void* stream) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(morton3d_invert_kernel, dim3(grid1d(m)), dim3(256), 0, (hipStream_t)stream, indices, m, coords);
    NGP_LAUNCH_CHECK();
    return 0;
}
int ngp_packbits(const float* density_grid, float threshold, int n_bytes, uint8_t* bitfield, void* stream) {
    if (n_bytes <= 0) return 0;
    hipLaunchKernelGGL(packbits_kernel, dim3(grid1d(n_bytes)), dim3(256), 0, (hipStream_t)stream, (const float4*)density_grid,
                       threshold, n_bytes, bitfield);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
