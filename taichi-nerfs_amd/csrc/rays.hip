// rays.hip -- camera-ray generation and training-batch sampling on the device.
//
// Replaces, for the training loop of the reference, datasets/ray_utils.py:51-80 (get_rays: a [N,1,3] x [N,3,3] batched matmul
// + an expand) and the three fancy-index gathers of datasets/base.py:34-61 (rays[img, pix], poses[img], directions[pix]):
// one lane per ray, 48-byte pose read (one per ray, or one broadcast pose for an eval image), three 12-byte writes.
// rays_d[i] = (d0 * R[i][0] + d1 * R[i][1]) + d2 * R[i][2], f32, no contraction (compiled with -ffp-contract=off): the
// summation order of the reference's bmm is library-defined, so parity against it is a 1-ulp tolerance, not bit equality.
#include "ngp_device.h"

namespace ngp {

__device__ __forceinline__ void ray_from_pose(const float* __restrict__ c2w, float d0, float d1, float d2, float* __restrict__ o,
                                              float* __restrict__ d) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r0 = c2w[4 * i], r1 = c2w[4 * i + 1], r2 = c2w[4 * i + 2];
        d[i] = (d0 * r0 + d1 * r1) + d2 * r2;
        o[i] = c2w[4 * i + 3];
    }
}

__global__ void __launch_bounds__(256) get_rays_kernel(const float* __restrict__ directions, const float* __restrict__ poses,
                                                       int pose_stride, int n, float* __restrict__ rays_o,
                                                       float* __restrict__ rays_d) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const float* dir = directions + 3 * (size_t)k;
        ray_from_pose(poses + (size_t)pose_stride * k, dir[0], dir[1], dir[2], rays_o + 3 * (size_t)k, rays_d + 3 * (size_t)k);
    }
}

__global__ void __launch_bounds__(256) sample_rays_kernel(const float* __restrict__ poses, const float* __restrict__ directions,
                                                          const float* __restrict__ rays, int ray_c, long long hw,
                                                          const int64_t* __restrict__ img_idx, long long img0,
                                                          const int64_t* __restrict__ pix_idx, int n, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, float* __restrict__ rgb) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const long long img = img_idx ? img_idx[k] : img0, pix = pix_idx[k];
        const float* dir = directions + 3 * pix;
        ray_from_pose(poses + 12 * img, dir[0], dir[1], dir[2], rays_o + 3 * (size_t)k, rays_d + 3 * (size_t)k);
        if (rgb) {
            const float* px = rays + ((size_t)img * hw + pix) * ray_c;
            rgb[3 * (size_t)k] = px[0]; rgb[3 * (size_t)k + 1] = px[1]; rgb[3 * (size_t)k + 2] = px[2];
        }
    }
}

// up to three equal-length float copies in one launch (batch -> the static buffers a captured training step reads)
__global__ void __launch_bounds__(256) stage_batch_kernel(const float* __restrict__ a, float* __restrict__ da,
                                                          const float* __restrict__ b, float* __restrict__ db,
                                                          const float* __restrict__ c, float* __restrict__ dc, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (a) da[i] = a[i];
        if (b) db[i] = b[i];
        if (c) dc[i] = c[i];
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_get_rays(const float* directions, const float* poses, int per_ray_pose, int n, float* rays_o, float* rays_d, void* stream) {
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(get_rays_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, directions, poses, per_ray_pose ? 12 : 0, n,
                       rays_o, rays_d);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_sample_rays(const float* poses, const float* directions, const float* rays, int ray_c, long long hw, const int64_t* img_idx,
                    long long img0, const int64_t* pix_idx, int n, float* rays_o, float* rays_d, float* rgb, void* stream) {
    if (n <= 0) return 0;
    if (rgb && (!rays || ray_c < 3)) return -1;
    int blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sample_rays_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, poses, directions, rays, ray_c, hw, img_idx,
                       img0, pix_idx, n, rays_o, rays_d, rgb);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_stage_batch(const float* a, float* da, const float* b, float* db, const float* c, float* dc, int n, void* stream) {
    if (n <= 0) return 0;
    if ((a && !da) || (b && !db) || (c && !dc)) return -1;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(stage_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, da, b, db, c, dc, n);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
