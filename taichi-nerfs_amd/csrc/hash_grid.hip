// hash_grid.hip -- multiresolution hash-grid encoding (fwd gather / bwd scatter-add) for gfx950.
//
// Replaces modules/hash_encoder.py:89-143 (+ its Taichi-autodiff backward, :269) and
// modules/hash_encoder_half.py:112-213 of the reference.
//
// Layout: table is the reference's flat [entries, F] array (level l occupies entries
// [offset_l, offset_l + size_l)); out/dout are [n, L*F], level-major, feature-minor.
// Mapping: one lane per (sample, level) with the level index fastest, so a wave covers 64/L consecutive
// samples and writes 64*F*4 contiguous bytes; the 8 corner loads of a lane are issued back to back
// (8 independent 8-byte gathers in flight per lane).  The level table sits in LDS.
#include "ngp_device.h"
#include <hip/hip_fp16.h>

namespace ngp {

struct LevelLDS {
    float scale[NGP_MAX_LEVELS];
    uint32_t res[NGP_MAX_LEVELS];
    uint32_t size[NGP_MAX_LEVELS];
    uint32_t offset[NGP_MAX_LEVELS];
    uint32_t mode[NGP_MAX_LEVELS];   // 0 dense (conditional subtract), 1 hashed pow2 (mask), 2 generic modulo
};

__device__ __forceinline__ void load_levels(const ngp_hash_levels& lv, LevelLDS& s) {
    int t = threadIdx.x;
    if (t < NGP_MAX_LEVELS) {
        s.scale[t] = lv.scale[t];
        s.res[t] = lv.resolution[t];
        uint32_t sz = lv.map_size[t];
        s.size[t] = sz;
        s.offset[t] = lv.offset[t];
        uint32_t mode;
        if (t < lv.begin_fast_hash_level) {
            // dense level: idx <= res^3 + res^2 + res < 2*size whenever size >= res^3, so one conditional
            // subtract equals `% size` (hash_encoder.py:71); anything else falls back to the real modulo.
            uint64_t r = lv.resolution[t];
            mode = ((uint64_t)sz >= r * r * r && r >= 2) ? 0u : 2u;
        } else {
            mode = (sz != 0 && (sz & (sz - 1)) == 0) ? 1u : 2u;
        }
        s.mode[t] = mode;
    }
    __syncthreads();
}

struct Corners {
    uint32_t idx[8];
    float w[8];
};

// hash_encoder.py:100-137; HALF_CELL applies hash_encoder_half.py:133 (cell cast to f16 before the subtract)
template <bool HALF_CELL>
__device__ __forceinline__ void corners(const LevelLDS& L, int level, int bfhl, float x, float y, float z, Corners& c) {
    const float scale = L.scale[level];
    const uint32_t res = L.res[level], size = L.size[level], mode = L.mode[level];
    float pos[3] = {x * scale + 0.5f, y * scale + 0.5f, z * scale + 0.5f};
    uint32_t cell[3];
    float fr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cell[k] = f2u_sat(floorf(pos[k]));
        float cf = (float)cell[k];
        if (HALF_CELL) cf = __half2float(__float2half_rn(cf));
        fr[k] = pos[k] - cf;
    }
    const bool dense = level < bfhl;
    const uint32_t res2 = res * res;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        float w = 1.0f;
        uint32_t g[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if ((ci & (1 << d)) == 0) { g[d] = cell[d]; w *= 1.0f - fr[d]; }
            else { g[d] = cell[d] + 1u; w *= fr[d]; }
        }
        uint32_t h = dense ? (g[0] + g[1] * res + g[2] * res2)                     // under_hash :53-60
                           : (g[0] ^ (g[1] * 2654435761u) ^ (g[2] * 805459861u));  // fast_hash :43-51
        if (mode == 1u) h &= (size - 1u);
        else if (mode == 0u) { if (h >= size) { h -= size; if (h >= size) h %= size; } }
        else h = h % size;
        c.idx[ci] = L.offset[level] + h;
        c.w[ci] = w;
    }
}

// ---- fp32 forward -----------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(256) hash_fwd_f32_kernel(const float* __restrict__ xyzs, const float* __restrict__ table,
                                                           ngp_hash_levels lv, int n, float* __restrict__ out) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const int nl = lv.n_levels;
    const long long total = (long long)n * nl;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(gid / nl), level = (int)(gid - (long long)i * nl);
        const float x = xyzs[3 * (size_t)i], y = xyzs[3 * (size_t)i + 1], z = xyzs[3 * (size_t)i + 2];
        Corners c;
        corners<false>(L, level, lv.begin_fast_hash_level, x, y, z, c);
        float v[8][F];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float* p = table + (size_t)c.idx[ci] * F;
            if constexpr (F == 2) { float2 t = *reinterpret_cast<const float2*>(p); v[ci][0] = t.x; v[ci][1] = t.y; }
            else if constexpr (F == 4) { float4 t = *reinterpret_cast<const float4*>(p); v[ci][0] = t.x; v[ci][1] = t.y; v[ci][2] = t.z; v[ci][3] = t.w; }
            else {
#pragma unroll
                for (int f = 0; f < F; ++f) v[ci][f] = p[f];
            }
        }
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += c.w[ci] * v[ci][f];               // :139-140 (mul then add)
        float* o = out + (size_t)gid * F;
        if constexpr (F == 2) *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[1]);
        else if constexpr (F == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else {
#pragma unroll
            for (int f = 0; f < F; ++f) o[f] = acc[f];
        }
    }
}

// ---- fp32 backward: dtable[idx*F+f] += w * dout -----------------------------------------------------
template <int F>
__global__ void __launch_bounds__(256) hash_bwd_f32_kernel(const float* __restrict__ xyzs, const float* __restrict__ dout,
                                                           ngp_hash_levels lv, int n, float* __restrict__ dtable) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const int nl = lv.n_levels;
    const long long total = (long long)n * nl;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(gid / nl), level = (int)(gid - (long long)i * nl);
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = dout[(size_t)gid * F + f]; any |= (g[f] != 0.0f); }
        if (!any) continue;            // samples behind early termination carry exact-zero gradients
        const float x = xyzs[3 * (size_t)i], y = xyzs[3 * (size_t)i + 1], z = xyzs[3 * (size_t)i + 2];
        Corners c;
        corners<false>(L, level, lv.begin_fast_hash_level, x, y, z, c);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int f = 0; f < F; ++f) unsafeAtomicAdd(dtable + (size_t)c.idx[ci] * F + f, c.w[ci] * g[f]);
    }
}

// ---- half2 forward (hash_encoder_half.py:112-161): f16 table, f16 accumulate ---------------------------
__global__ void __launch_bounds__(256) hash_fwd_f16_kernel(const float* __restrict__ xyzs, const __half2* __restrict__ table,
                                                           ngp_hash_levels lv, int n, __half2* __restrict__ out) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const int nl = lv.n_levels;
    const long long total = (long long)n * nl;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(gid / nl), level = (int)(gid - (long long)i * nl);
        const float x = xyzs[3 * (size_t)i], y = xyzs[3 * (size_t)i + 1], z = xyzs[3 * (size_t)i + 2];
        Corners c;
        corners<true>(L, level, lv.begin_fast_hash_level, x, y, z, c);
        __half2 v[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) v[ci] = table[c.idx[ci]];
        __half2 acc = __floats2half2_rn(0.0f, 0.0f);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            float2 tv = __half22float2(v[ci]);
            __half2 term = __floats2half2_rn(c.w[ci] * tv.x, c.w[ci] * tv.y);       // cast(w*table, f16) :159
            acc = __hadd2(acc, term);
        }
        out[gid] = acc;
    }
}

// ---- half2 backward (hash_encoder_half.py:164-213): one packed f16x2 atomic per corner -------------------
__global__ void __launch_bounds__(256) hash_bwd_f16_kernel(const float* __restrict__ xyzs, const __half2* __restrict__ dout,
                                                           ngp_hash_levels lv, int n, __half2* __restrict__ dtable) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const int nl = lv.n_levels;
    const long long total = (long long)n * nl;
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(gid / nl), level = (int)(gid - (long long)i * nl);
        const float2 g = __half22float2(dout[gid]);
        if (g.x == 0.0f && g.y == 0.0f) continue;                                   // :210
        const float x = xyzs[3 * (size_t)i], y = xyzs[3 * (size_t)i + 1], z = xyzs[3 * (size_t)i + 2];
        Corners c;
        corners<true>(L, level, lv.begin_fast_hash_level, x, y, z, c);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            half2v val;
            val.x = (_Float16)(c.w[ci] * g.x);
            val.y = (_Float16)(c.w[ci] * g.y);
            if (val.x == (_Float16)0 && val.y == (_Float16)0) continue;             // :212
            __builtin_amdgcn_global_atomic_fadd_v2f16(
                (__attribute__((address_space(1))) half2v*)(dtable + c.idx[ci]), val);   // global_atomic_pk_add_f16
        }
    }
}

inline int grid_for(long long work, int block) {
    long long b = (work + block - 1) / block;
    const long long cap = 256LL * 16;      // 256 CUs x 16 blocks, grid-stride beyond that
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_abi_version(void) { return NGP_ABI_VERSION; }

int ngp_hash_fwd_f32(const float* xyzs, const float* table, const ngp_hash_levels* lv, int n, float* out, void* stream) {
    if (n <= 0) return 0;
    if (lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n * lv->n_levels, 256);
    hipStream_t s = (hipStream_t)stream;
    switch (lv->n_features) {
        case 1: hipLaunchKernelGGL(hash_fwd_f32_kernel<1>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n, out); break;
        case 2: hipLaunchKernelGGL(hash_fwd_f32_kernel<2>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n, out); break;
        case 4: hipLaunchKernelGGL(hash_fwd_f32_kernel<4>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n, out); break;
        case 8: hipLaunchKernelGGL(hash_fwd_f32_kernel<8>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n, out); break;
        default: return -1;
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_f32(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n, float* dtable, void* stream) {
    if (n <= 0) return 0;
    if (lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n * lv->n_levels, 256);
    hipStream_t s = (hipStream_t)stream;
    switch (lv->n_features) {
        case 1: hipLaunchKernelGGL(hash_bwd_f32_kernel<1>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n, dtable); break;
        case 2: hipLaunchKernelGGL(hash_bwd_f32_kernel<2>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n, dtable); break;
        case 4: hipLaunchKernelGGL(hash_bwd_f32_kernel<4>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n, dtable); break;
        case 8: hipLaunchKernelGGL(hash_bwd_f32_kernel<8>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n, dtable); break;
        default: return -1;
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_fwd_f16(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n, uint16_t* out, void* stream) {
    if (n <= 0) return 0;
    if (lv->n_features != 2 || lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n * lv->n_levels, 256);
    hipLaunchKernelGGL(hash_fwd_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, xyzs, (const __half2*)table, *lv, n,
                       (__half2*)out);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_f16(const float* xyzs, const uint16_t* dout, const ngp_hash_levels* lv, int n, uint16_t* dtable, void* stream) {
    if (n <= 0) return 0;
    if (lv->n_features != 2 || lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n * lv->n_levels, 256);
    hipLaunchKernelGGL(hash_bwd_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, xyzs, (const __half2*)dout, *lv, n,
                       (__half2*)dtable);
    NGP_LAUNCH_CHECK();
    return 0;
}

// Host-only level-table builder: modules/hash_encoder.py:183-205 + modules/utils.py:19-42.
int ngp_hash_levels_init(ngp_hash_levels* lv, double max_params, int levels, double base_res, double max_res, int features) {
    if (!lv || levels < 1 || levels > NGP_MAX_LEVELS) return -1;
    *lv = ngp_hash_levels{};
    const double log_b = (levels > 1) ? log(max_res / base_res) / (double)(levels - 1) : 0.0;
    unsigned long long offset = 0;
    int bfhl = levels;
    for (int i = 0; i < levels; ++i) {
        const double resolution = ceil(base_res * exp((double)i * log_b) - 1.0) + 1.0;
        const double full = resolution * resolution * resolution;
        const double aligned = (double)((long long)((full + 7.0) / 8.0) * 8);
        const double size = max_params < aligned ? max_params : aligned;
        lv->offset[i] = (uint32_t)offset;
        lv->map_size[i] = (uint32_t)size;
        if (full > size && bfhl == levels) bfhl = i;
        offset += (unsigned long long)size;
        const float sc = (float)base_res * expf((float)i * (float)log_b) - 1.0f;
        lv->scale[i] = sc;
        lv->resolution[i] = (uint32_t)ceilf(sc) + 1u;
    }
    lv->n_levels = levels;
    lv->n_features = features;
    lv->begin_fast_hash_level = bfhl;
    lv->total_entries = (int32_t)offset;
    return 0;
}

}  // extern "C"
