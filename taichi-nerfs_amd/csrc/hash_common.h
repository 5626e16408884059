// hash_common.h -- level table in LDS + position normalisation shared by hash_grid.hip and hash_bwd_lds.hip.
#pragma once
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

struct XyzNorm {            // optional fused (x - lo) / (hi - lo) of reference networks.py:144 (same two f32 ops)
    int enabled;
    float lo, hi;
};
__device__ __forceinline__ float norm01(const XyzNorm& nm, float v) { return nm.enabled ? (v - nm.lo) / (nm.hi - nm.lo) : v; }


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

}  // namespace ngp
