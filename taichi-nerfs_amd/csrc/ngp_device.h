// ngp_device.h -- device-side helpers shared by the gfx950 kernels of libngp_hip.so.
//
// Everything on the bit-exact paths (ray orbit, Morton / bitfield indexing, hash-grid corner indices and
// trilinear weights) is written as separate IEEE binary32 multiplies and adds; the library is compiled with
// -ffp-contract=off so the compiler never fuses them, and explicit fmaf() is used only where a kernel is
// tolerance-checked instead of bit-checked.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"
#include "../../include/ngp_hip_experimental.h"
#include <stdlib.h>
#include <string.h>

// NGP_EXPERIMENT="key=value;key=value": the ONE environment variable behind which every launch-shape / A-B knob of the library lives
// (round 6; until round 5 each had an NGP_* variable of its own).  Returns the value of `key` in a thread-local buffer, or NULL.
// The keys are listed, with what they do, in ngp_hip/experiment.py (KEYS) and INTEGRATION.md section 5; none is needed -- every default
// is the measured-fastest form.
static inline const char* ngp_experiment(const char* key) {
    const char* e = getenv("NGP_EXPERIMENT");
    if (!e || !key) return nullptr;
    static thread_local char buf[64];
    const size_t kl = strlen(key);
    while (*e) {
        while (*e == ';' || *e == ' ') ++e;
        const char* end = strchr(e, ';');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        if (len > kl && strncmp(e, key, kl) == 0 && e[kl] == '=') {
            size_t vl = len - kl - 1;
            if (vl >= sizeof(buf)) vl = sizeof(buf) - 1;
            memcpy(buf, e + kl + 1, vl);
            buf[vl] = 0;
            return buf;
        }
        e += len;
    }
    return nullptr;
}


#define NGP_WAVE 64

// One target, no dual paths: the kernels use gfx950 instructions and registers directly (v_cvt_u32_f32 saturation semantics,
// s_getreg HW_REG_XCC_ID, ds_read_b64_tr_b16, v_mfma_f32_16x16x32_f16, the 160 KB LDS).  Fail the build, not the run, elsewhere.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libngp_hip is written for gfx950 (MI355X / CDNA4): build with --offload-arch=gfx950"
#endif

#define NGP_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return -(int)e__;             \
    } while (0)

namespace ngp {

__device__ __forceinline__ uint32_t f2u_bits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f_bits(uint32_t u) { return __uint_as_float(u); }

// f32 -> u32 truncating, saturating cast: NaN / negative -> 0, >= 2^32 -> 0xffffffff.  That is exactly what the hardware's
// v_cvt_u32_f32 does; issuing it directly keeps the three-way C definition (two compares + branches per coordinate in every
// hash / march kernel) out of the instruction stream.
__device__ __forceinline__ uint32_t f2u_sat(float v) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// f32 -> f16 (round to nearest even) as ONE instruction the compiler cannot fold into the multiply that produced its operand.
// clang lowers `(_Float16)(a * b)` -- and __floats2half2_rn(a * b, ...) -- to v_fma_mixlo_f16: the exact product rounded ONCE to
// f16, even under -ffp-contract=off.  The reference's half2 encoder multiplies in f32 and then casts (hash_encoder_half.py:159,
// 205-212: two roundings); the two differ on ~1 product in 2000 by one f16 ulp (found by the bit-exact single-contribution rows
// of tests/golden/ref_hash_f16*.npz).  gfx9-family 16-bit VALU results zero the upper half of the destination register.
__device__ __forceinline__ uint32_t f16_bits_rn(float v) {
    uint32_t r;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(v));
    return r & 0xffffu;
}
__device__ __forceinline__ float f16_round(float v) {                     // through f16 and back
    float r;
    asm("v_cvt_f16_f32 %0, %1\n\tv_cvt_f32_f16 %0, %0" : "=v"(r) : "v"(v));
    return r;
}

// modules/utils.py:54-57
__device__ __forceinline__ float calc_dt(float t, float esf, float dt_min, float dt_max) {
    return fminf(dt_max, fmaxf(dt_min, t * esf));
}

// modules/utils.py:60-75 (differs from C frexp on exact powers of two)
__device__ __forceinline__ int frexp_bit(float x) {
    int exponent = 0;
    if (x != 0.0f) {
        uint32_t bits = f2u_bits(x);
        exponent = (int)((bits & 0x7f800000u) >> 23) - 127;
        float frac = u2f_bits((bits & 0x7fffffu) | 0x3f800000u);
        if (frac > 1.0f) exponent += 1;   // the frac < 0.5 branch of the reference can never fire
    }
    return exponent;
}

// modules/utils.py:95-107
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
// modules/utils.py:110-117
__device__ __forceinline__ int32_t morton3d_invert1(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return (int32_t)x;
}

__device__ __forceinline__ float fsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

// ---- wave64 primitives -------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (NGP_WAVE - 1)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, NGP_WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, NGP_WAVE);
    return v;
}
// inclusive scans across the 64 lanes (Kogge-Stone)
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int d = 1; d < NGP_WAVE; d <<= 1) {
        float o = __shfl_up(v, d, NGP_WAVE);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int d = 1; d < NGP_WAVE; d <<= 1) {
        float o = __shfl_up(v, d, NGP_WAVE);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ int wave_scan_add_i(int v, int lane) {
#pragma unroll
    for (int d = 1; d < NGP_WAVE; d <<= 1) {
        int o = __shfl_up(v, d, NGP_WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// ---- optimizer state layout (include/ngp_hip.h) and the dense Adam pass shared by optim.hip and mlp.hip ----------------
enum { SF_LOSS_SCALE = 0, SF_INV_SCALE = 1, SF_LR = 2, SF_BC1 = 3, SF_BC2_SQRT = 4, SF_LOSS = 5, SF_LOSS_ACC = 6 };
// Counter-based uniform in [0, 1) with float32's 24-bit resolution (what torch.rand gives): splitmix64 of (seed, index).  The march
// jitter of ray r in the step with seed s is rng_uniform(s, r): no noise tensor, no generator launch on the stream (reference:
// torch.rand_like, ray_march.py:138 -- any i.i.d. uniform does; tests/test_gpu_parity.py holds the distribution and the
// equivalence with an explicit noise vector of the same values).
__host__ __device__ __forceinline__ float rng_uniform(unsigned long long seed, unsigned int index) {
    unsigned long long z = seed + ((unsigned long long)index + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned int)(z >> 40) * (1.0f / 16777216.0f);
}

// ---- cross-block sum of the MLP backward's per-block weight-gradient slabs (csrc/mlp.hip: the kernels leave one [NGP_MLP_NW]
// slab of plain stores per block instead of 2.6 M same-address float atomics, 13 us of a 72 us launch).  A 1024-thread block sums
// 64 weights: thread (w = tid & 63, q = tid >> 6) takes every 16th slab -- at most 16 loads, all in flight at once: the pass is
// latency-, not bandwidth-bound (10 MB) --, the sixteen partial sums meet in LDS.  Used by the stand-alone reduction
// (ngp_mlp_dw_reduce) and by the trainer's prologue launch, which has to exist anyway.
constexpr int NGP_MLP_NW = 9408;                       // W1 | W2 | W3 | W4 | W5
constexpr int NGP_MLP_REDUCE_BLOCKS = (NGP_MLP_NW + 63) / 64;
constexpr int NGP_MLP_REDUCE_THREADS = 1024;
__device__ __forceinline__ void mlp_dw_reduce_block(const float* __restrict__ parts, int n_parts, float* __restrict__ dW, int block) {
    __shared__ float partial[15][64];
    const int lane = threadIdx.x & 63, w = block * 64 + lane, q = threadIdx.x >> 6;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int p = q + 16 * k;
        v[k] = (w < NGP_MLP_NW && p < n_parts) ? parts[(size_t)p * NGP_MLP_NW + w] : 0.0f;
    }
    float s = 0.0f;
    for (int p = q + 256; p < n_parts; p += 16) s += (w < NGP_MLP_NW) ? parts[(size_t)p * NGP_MLP_NW + w] : 0.0f;   // (n_parts > 256)
#pragma unroll
    for (int k = 0; k < 16; k += 4) s += (v[k] + v[k + 1]) + (v[k + 2] + v[k + 3]);
    if (q) partial[q - 1][lane] = s;
    __syncthreads();
    if (q == 0 && w < NGP_MLP_NW) {
#pragma unroll
        for (int k = 0; k < 15; ++k) s += partial[k][lane];
        dW[w] += s;
    }
}

enum { SI_ITER = 0, SI_OPT_STEP = 1, SI_GROWTH = 2, SI_FOUND_INF = 3, SI_SKIP = 4, SI_SKIPPED_TOTAL = 5 };

// The training step's scalar bookkeeping (reference train.py:197-201: GradScaler.step's skip decision, GradScaler.update,
// CosineAnnealingLR.step, Adam's bias corrections), one thread, on the state layout of include/ngp_hip.h.  sf / si may be LDS
// copies: the scatter-add of round 5 (hash_bwd_lds.hip) runs it per workgroup on a private copy and lets the last workgroup out
// publish the result, which removes the one-thread launch (6 us + a 4 us gap) from in front of it.
__device__ __forceinline__ void train_prologue_thread(float* __restrict__ sf, int32_t* __restrict__ si, float lr0, float eta_min, int t_max,
                                                      float beta1, float beta2, float growth, float backoff, int growth_interval) {
    const int iter = si[SI_ITER];
    const int found = si[SI_FOUND_INF];
    const float scale = sf[SF_LOSS_SCALE];
    sf[SF_INV_SCALE] = 1.0f / scale;                                         // GradScaler.unscale_
    si[SI_SKIP] = found ? 1 : 0;
    // CosineAnnealingLR closed form; scheduler.step() runs every iteration, skipped or not (train.py:201)
    const float c = cosf(3.14159265358979323846f * (float)(iter < t_max ? iter : t_max) / (float)t_max);
    sf[SF_LR] = eta_min + (lr0 - eta_min) * 0.5f * (1.0f + c);
    if (!found) {
        const int step = si[SI_OPT_STEP] + 1;
        si[SI_OPT_STEP] = step;
        sf[SF_BC1] = 1.0f - powf(beta1, (float)step);
        sf[SF_BC2_SQRT] = sqrtf(1.0f - powf(beta2, (float)step));
        int g = si[SI_GROWTH] + 1;                                           // GradScaler.update, growth branch
        if (g >= growth_interval) { sf[SF_LOSS_SCALE] = scale * growth; g = 0; }
        si[SI_GROWTH] = g;
    } else {
        sf[SF_LOSS_SCALE] = scale * backoff;                                 // GradScaler.update, backoff branch
        si[SI_GROWTH] = 0;
        si[SI_SKIPPED_TOTAL] += 1;
    }
    si[SI_FOUND_INF] = 0;
    si[SI_ITER] = iter + 1;
}

// round-to-nearest-even f32 -> bf16 (what torch's .bfloat16() does); NaN stays NaN
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// round-to-nearest-even f32 -> f16 bits (the per-forward cast of hash_encoder_half.py:367)
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }

// torch.optim.Adam(eps) arithmetic on float4s, grid-stride over `n_blocks` blocks of which this is `block`; unscales the
// gradient on the fly and zero-fills it.
// SHADOW: also refresh a 16-bit storage copy of the parameters (1 = bf16, 2 = f16).  GRAD16: the gradient buffer is f16 (the
// half2 encoder's, hash_encoder_half.py:350-358) and is widened to f32 here, like autograd does for the fp32 master parameter.
template <int SHADOW, bool GRAD16 = false>
__device__ __forceinline__ void adam_table_pass(float4* __restrict__ p, void* __restrict__ gv, float4* __restrict__ m,
                                                float4* __restrict__ v, long n4, const float* __restrict__ sf,
                                                const int32_t* __restrict__ si, float beta1, float beta2, float eps,
                                                uint2* __restrict__ shadow, long block, long n_blocks) {
    const bool skip = si[SI_SKIP] != 0;
    const float inv_scale = sf[SF_INV_SCALE], step_size = sf[SF_LR] / sf[SF_BC1], bc2_sqrt = sf[SF_BC2_SQRT];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* g32 = reinterpret_cast<float4*>(gv);
    uint2* g16 = reinterpret_cast<uint2*>(gv);
    for (long i = block * blockDim.x + threadIdx.x; i < n4; i += n_blocks * blockDim.x) {
        float4 gi;
        if constexpr (GRAD16) {
            const uint2 r = g16[i];
            const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&r.x)), b = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
            gi = make_float4(a.x, a.y, b.x, b.y);
        } else {
            gi = g32[i];
        }
        // the gradient slot is cleared for the next step -- only where it holds something: most float4 groups of the fine levels
        // receive no gradient in a given step, and a store of zeros over zeros is 16 bytes of HBM write traffic per group
        const bool g_any = !(gi.x == 0.f && gi.y == 0.f && gi.z == 0.f && gi.w == 0.f);
        if (skip) {
            if (g_any) { if constexpr (GRAD16) g16[i] = make_uint2(0u, 0u); else g32[i] = zero; }
            continue;
        }
        float4 mi = m[i], vi = v[i];
        // an entry that never received a gradient (g = m = v = 0) is a fixed point of Adam: m' = v' = 0 and the update is
        // lr * 0 / (0 + eps) = 0 exactly -- skip its parameter read and all four writes (hashed levels of a sparse scene
        // leave a large part of the table untouched for the whole run)
        if (!g_any && mi.x == 0.f && mi.y == 0.f && mi.z == 0.f && mi.w == 0.f && vi.x == 0.f && vi.y == 0.f && vi.z == 0.f && vi.w == 0.f)
            continue;
        float4 pi = p[i];
#define NGP_ADAM1(c)                                                          \
        {                                                                     \
            const float gr = gi.c * inv_scale;                                \
            mi.c = mi.c + (gr - mi.c) * (1.0f - beta1);                       \
            vi.c = vi.c * beta2 + gr * gr * (1.0f - beta2);                   \
            const float denom = sqrtf(vi.c) / bc2_sqrt + eps;                 \
            pi.c = pi.c - step_size * (mi.c / denom);                         \
        }
        NGP_ADAM1(x) NGP_ADAM1(y) NGP_ADAM1(z) NGP_ADAM1(w)
#undef NGP_ADAM1
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (g_any) { if constexpr (GRAD16) g16[i] = make_uint2(0u, 0u); else g32[i] = zero; }
        if constexpr (SHADOW == 1)
            shadow[i] = make_uint2(f32_to_bf16_bits(pi.x) | (f32_to_bf16_bits(pi.y) << 16),
                                   f32_to_bf16_bits(pi.z) | (f32_to_bf16_bits(pi.w) << 16));
        if constexpr (SHADOW == 2)
            shadow[i] = make_uint2(f32_to_f16_bits(pi.x) | (f32_to_f16_bits(pi.y) << 16),
                                   f32_to_f16_bits(pi.z) | (f32_to_f16_bits(pi.w) << 16));
    }
}

}  // namespace ngp
