// hash_common.h -- level table in LDS + position normalisation shared by hash_grid.hip and hash_bwd_lds.hip.
#pragma once
#include "ngp_device.h"

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


}  // namespace ngp
