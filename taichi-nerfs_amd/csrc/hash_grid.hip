// hash_grid.hip -- multiresolution hash-grid encoding (fwd gather / bwd scatter-add) for gfx950.
//
// Replaces modules/hash_encoder.py:89-143 (+ its Taichi-autodiff backward, :269) and
// modules/hash_encoder_half.py:112-213 of the reference.
//
// Layout: table is the reference's flat [entries, F] array (level l occupies entries
// [offset_l, offset_l + size_l)); out/dout are [n, L*F], level-major, feature-minor -- or, on the fused path (L = 16,
// F = 2), eight pair-major planes [8][n_max][4] (plane p = levels p and 15 - p).
// Kernels, generic to specialised:
//   hash_fwd_f32_kernel<F>        one lane per (sample, level), level fastest; 8 independent gathers in flight per lane
//   hash_fwd_f32_xcd_kernel<MODE> block b encodes level pair (b % 8, 15 - b % 8): every XCD's L2 keeps its slice of the table;
//                                 MODE selects the fp32 table, a bf16 storage copy, or the half2 encoder's f16 arithmetic
//   hash_bwd_f32_kernel<F>        generic scatter-add, one lane per (sample, level)
//   hash_bwd_f32x2_kernel         F = 2: lane quads on one 64-byte line + run merging (the atomic line-request rate is the bound)
//   hash_fwd/bwd_f16_kernel       the half2 encoder as the reference writes it; hash_bwd_f16x2_kernel its fused-path form
// The backward kernels optionally run over a compacted list of live samples (live_idx).  The level table sits in LDS.
#include "ngp_device.h"
#include "hash_common.h"
#include <hip/hip_fp16.h>

namespace ngp {

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

// The same corners for ONE level whose constants the caller holds in registers, without a branch: both index forms are one
// three-operand instruction per corner (v_add3 / v_xor3) behind per-axis terms ((c + 1) * k = c * k + k mod 2^32), and `% size` is
// `& (size - 1)` on a power-of-two hashed level or one conditional subtract (min(h, h - size), unsigned) on a dense one -- exactly
// `corners` above, but with c.idx RELATIVE to the level's first entry (the caller folds the offset into its base pointer); a lane
// whose level needs the real modulo (mode 2, or a dense index >= 2 * size) reports it in the return value and the caller redoes
// that lane's eight indices with `%` under ONE rarely taken branch.  Straight-line code keeps the
// eight gathers of an iteration -- and the next iteration's position request -- in one basic block.
struct LevelRegs {
    float scale;
    uint32_t res, size, mode, offset;
    bool dense;
    // one form for both cheap cases: r = min(h & mask, (h & mask) - wrap), unsigned.  Power-of-two hashed level: mask = size - 1 and
    // wrap = 2^31 (never taken: h & mask < 2^31).  Dense level: mask = all ones, wrap = size (the conditional subtract).
    uint32_t mask, wrap;
    __device__ __forceinline__ void derive() {
        mask = mode == 1u ? size - 1u : 0xffffffffu;
        wrap = mode == 1u ? 0x80000000u : size;
    }
};
template <bool HALF_CELL>
__device__ __forceinline__ bool corners_flat(const LevelRegs& lr, float x, float y, float z, Corners& c, uint32_t h_raw[8]) {
    float pos[3] = {x * lr.scale + 0.5f, y * lr.scale + 0.5f, z * lr.scale + 0.5f};
    uint32_t cell[3];
    float fr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cell[k] = f2u_sat(floorf(pos[k]));
        float cf = (float)cell[k];
        if (HALF_CELL) cf = __half2float(__float2half_rn(cf));
        fr[k] = pos[k] - cf;
    }
    const uint32_t ky = lr.dense ? lr.res : 2654435761u, kz = lr.dense ? lr.res * lr.res : 805459861u;
    const uint32_t ax[2] = {cell[0], cell[0] + 1u};
    const uint32_t ay[2] = {cell[1] * ky, cell[1] * ky + ky};
    const uint32_t az[2] = {cell[2] * kz, cell[2] * kz + kz};
    uint32_t worst = 0u;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
        float w = 1.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) w *= (ci & (1 << d)) ? fr[d] : 1.0f - fr[d];
        const uint32_t gx = ax[ci & 1], gy = ay[(ci >> 1) & 1], gz = az[(ci >> 2) & 1];
        const uint32_t h = lr.dense ? gx + gy + gz : gx ^ gy ^ gz;                 // under_hash :53-60 / fast_hash :43-51
        h_raw[ci] = h;
        const uint32_t hm = h & lr.mask, r = min(hm, hm - lr.wrap);
        worst = max(worst, r);
        c.idx[ci] = r;                                                             // relative to the level's first entry
        c.w[ci] = w;
    }
    return lr.mode != 1u && worst >= lr.size;
}

// BF16 = true: the table is stored as bf16 pairs (uint32 per entry, F = 2 only); bf16 -> f32 is exact, the interpolation and
// the output stay f32, so the result equals the f32 kernel run on the bf16-rounded table, bit for bit.
__device__ __forceinline__ float2 bf16x2_to_f32(uint32_t u) {
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}

template <int F, bool BF16 = false>
__global__ void __launch_bounds__(256) hash_fwd_f32_kernel(const float* __restrict__ xyzs, const float* __restrict__ table,
                                                           ngp_hash_levels lv, int n, const int32_t* __restrict__ n_dev,
                                                           XyzNorm nm, float* __restrict__ out) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    if (n_dev) n = min(n, *n_dev);
    const int nl = lv.n_levels;
    const long long total = (long long)n * nl;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(gid / nl), level = (int)(gid - (long long)i * nl);
        const float x = norm01(nm, xyzs[3 * (size_t)i]), y = norm01(nm, xyzs[3 * (size_t)i + 1]),
                    z = norm01(nm, xyzs[3 * (size_t)i + 2]);
        Corners c;
        corners<false>(L, level, lv.begin_fast_hash_level, x, y, z, c);
        float v[8][F];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const float* p = table + (size_t)c.idx[ci] * F;
            if constexpr (BF16) { float2 t = bf16x2_to_f32(reinterpret_cast<const uint32_t*>(table)[c.idx[ci]]); v[ci][0] = t.x; v[ci][1] = t.y; }
            else if constexpr (F == 2) { float2 t = *reinterpret_cast<const float2*>(p); v[ci][0] = t.x; v[ci][1] = t.y; }
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

// ---- fp32 forward, XCD-partitioned (L = 16, F = 2) --------------------------------------------------------
// Measured (profiles/microbench/atomics3.hip): 8-byte gathers confined to a 4 MB slice per XCD run 2.6x faster than the
// same gathers spread over the whole table, because each XCD's 4 MiB L2 then holds its slice instead of 1/11 of 45.7 MB.
// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8; a speed assumption only), so block b
// encodes the level pair (p, 15-p), p = b % 8: coarse + fine levels paired so every XCD owns a similar byte count.
// Lanes (2s, 2s+1) of a block handle the two levels of sample s.  Output stays in the reference layout [n, 32].
// enc_pairs = 1 writes the PAIR-MAJOR layout out[(pair * n_max + i) * 4 + which * 2 + f] (pair = min(l, 15-l), which = l >= 8):
// every block then stores 16 contiguous bytes per sample instead of two 8-byte pieces of a 128-byte row shared with the
// other seven XCDs (another 1.5x, same microbenchmark); the fused MLP kernels and the scatter-add consume that layout.
// MODE 0: f32 table.  MODE 1: bf16 storage copy, f32 arithmetic.  MODE 2: the half2 encoder's arithmetic (hash_encoder_half.py:
// 112-161: f16 table, cell cast to f16 before the subtract, every w * table term rounded to f16, f16 accumulation) with the
// result widened to f32 (exact) so the consumers of the fused path stay the same.
// Round 4: the loop body is one basic block.  The position of the NEXT iteration is requested (index clamped, so the request is
// unconditional) before this iteration's index arithmetic, the level's constants live in registers, and the corners come from
// corners_flat: an iteration used to be four dependent memory round trips (x, y, z each behind a branch of norm01, then the
// gathers) and ~60 scalar branches; it is now the gathers' round trip alone.  V1 = the round-1..3 loop (NGP_EXPERIMENT hash_fwd_v1=1), kept
// for the A/B in profiles/r04_hash_fwd_loop_experiment.txt.
// LIST (round 5, the chunked forward of FusedTrainer on scenes whose rays terminate long before their marched samples end): the
// launch encodes the samples list[0 .. *n_dev) -- position j reads xyzs[list[j]] and writes row list[j] of `out` -- instead of
// the first *n_dev rows.  The list entry of the iteration after next is requested where the next position is, so the indirection
// adds no round trip to the loop.
template <int MODE, bool V1 = false, bool LIST = false>
__global__ void __launch_bounds__(256) hash_fwd_f32_xcd_kernel(const float* __restrict__ xyzs, const float* __restrict__ table,
                                                               ngp_hash_levels lv, int n, const int32_t* __restrict__ n_dev,
                                                               XyzNorm nm, int enc_pairs, float* __restrict__ out,
                                                               const int32_t* __restrict__ list = nullptr) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const size_t plane = (size_t)n;                       // pair-major plane stride = buffer capacity
    if (n_dev) n = min(n, *n_dev);
    if (n <= 0) return;
    const int pair = blockIdx.x & 7, which = threadIdx.x & 1;
    const int level = which ? 15 - pair : pair;
    const int tiles = gridDim.x >> 3;
    // V1: c.idx = entry in the table (64-bit address per corner).  Otherwise c.idx = entry in the level and `off` = the level's
    // first BYTE: a 32-bit byte offset on the uniform table pointer (global_load ... v_off, s[base]) -- one VGPR per corner
    // instead of two; the host entry point takes this form only for tables below 4 GB.
    auto gather = [&](uint32_t off, const Corners& c, float2 v[8]) {
        const char* tb = reinterpret_cast<const char*>(table);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            if constexpr (V1) {
                if constexpr (MODE == 1) v[ci] = bf16x2_to_f32(reinterpret_cast<const uint32_t*>(table)[c.idx[ci]]);
                else if constexpr (MODE == 2) v[ci] = __half22float2(reinterpret_cast<const __half2*>(table)[c.idx[ci]]);
                else v[ci] = *reinterpret_cast<const float2*>(table + (size_t)c.idx[ci] * 2);
            } else {
                if constexpr (MODE == 1) v[ci] = bf16x2_to_f32(*reinterpret_cast<const uint32_t*>(tb + (c.idx[ci] * 4u + off)));
                else if constexpr (MODE == 2) v[ci] = __half22float2(*reinterpret_cast<const __half2*>(tb + (c.idx[ci] * 4u + off)));
                else v[ci] = *reinterpret_cast<const float2*>(tb + (c.idx[ci] * 8u + off));
            }
        }
    };
    auto blend = [&](const Corners& c, const float2 v[8]) {
        float a0 = 0.0f, a1 = 0.0f;
        if constexpr (MODE == 2) {
            __half2 acc = __floats2half2_rn(0.0f, 0.0f);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) acc = __hadd2(acc, __floats2half2_rn(c.w[ci] * v[ci].x, c.w[ci] * v[ci].y));   // :159
            const float2 r = __half22float2(acc);
            a0 = r.x; a1 = r.y;
        } else {
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) { a0 += c.w[ci] * v[ci].x; a1 += c.w[ci] * v[ci].y; }   // same order as the generic kernel
        }
        return make_float2(a0, a1);
    };
    auto out_at = [&](int i) { return reinterpret_cast<float2*>(enc_pairs ? out + ((size_t)pair * plane + i) * 4 + which * 2 : out + (size_t)i * 32 + level * 2); };
    if constexpr (V1) {
        for (int i = (blockIdx.x >> 3) * 128 + (threadIdx.x >> 1); i < n; i += tiles * 128) {
            const float x = norm01(nm, xyzs[3 * (size_t)i]), y = norm01(nm, xyzs[3 * (size_t)i + 1]),
                        z = norm01(nm, xyzs[3 * (size_t)i + 2]);
            Corners c;
            corners<MODE == 2>(L, level, lv.begin_fast_hash_level, x, y, z, c);
            float2 v[8];
            gather(0u, c, v);
            *out_at(i) = blend(c, v);
        }
    } else {
        LevelRegs lr = {L.scale[level], L.res[level], L.size[level], L.mode[level], L.offset[level], level < lv.begin_fast_hash_level};
        lr.derive();
        const uint32_t level_off = lr.offset * (MODE == 0 ? 8u : 4u);
        const int stride = tiles * 128;
        int i = (blockIdx.x >> 3) * 128 + (threadIdx.x >> 1);
        // norm01 = (v - lo) / (hi - lo), the reference's two f32 ops (networks.py:144).  Where hi - lo is a power of two (every scale
        // the reference ships: 0.5, 1, 2, ... 16), x / 2^k and x * 2^-k are the same correctly rounded value: one multiply
        // instead of the ~10-instruction IEEE division, three times per sample and level.
        // No normalisation is the same multiply with lo = 0 and 2^-k = 1 (v - 0 and v * 1 are exact).
        const float den = nm.enabled ? nm.hi - nm.lo : 1.0f, lo = nm.enabled ? nm.lo : 0.0f;
        const uint32_t den_bits = __float_as_uint(den);
        const bool den_pow2 = (den_bits & 0x807fffffu) == 0u && den_bits >= 0x20000000u && den_bits <= 0x5f800000u;   // 2^-63 .. 2^64
        const float inv_den = den_pow2 ? 1.0f / den : 0.0f;
        float2* o_prev = nullptr;
        float2 r_prev = make_float2(0.0f, 0.0f);
        float p[3];
        int s_cur = LIST ? list[min(i, n - 1)] : min(i, n - 1);                       // the sample this iteration encodes ...
        int s_next = LIST ? list[min(i + stride, n - 1)] : 0;                         // ... and the next one's (LIST only)
        {
            const float* q = xyzs + 3 * (size_t)s_cur;
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        }
        for (; i < n; i += stride) {
            float xyz[3];
            if (den_pow2) {
#pragma unroll
                for (int k = 0; k < 3; ++k) xyz[k] = (p[k] - lo) * inv_den;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) xyz[k] = (p[k] - lo) / den;
            }
            int s_next2 = 0;
            if constexpr (LIST) s_next2 = list[min(i + 2 * stride, n - 1)];
            {
                const float* q = xyzs + 3 * (size_t)(LIST ? s_next : min(i + stride, n - 1));
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            }
            Corners c;
            uint32_t h_raw[8];
            if (corners_flat<MODE == 2>(lr, xyz[0], xyz[1], xyz[2], c, h_raw)) {
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) c.idx[ci] = h_raw[ci] % lr.size;
            }
#ifdef NGP_HASH_FWD_DIAG
            // timing experiment (profiles/microbench/encoder_ab.py NGP_EXPERIMENT hash_fwd_free_levels): the gathers of the levels in the mask
            // all read entry 0 of the level -- one line, always in the vector L1 -- i.e. what staging those levels' tables in LDS
            // could at best buy; results are wrong by construction
            if ((nm.enabled >> (8 + level)) & 1) {
#pragma unroll
                for (int ci = 0; ci < 8; ++ci) c.idx[ci] = 0u;
            }
#endif
            float2 v[8];
            gather(level_off, c, v);
            // the PREVIOUS iteration's result is written here, underneath this iteration's gathers: gfx950 counts loads and stores
            // in one counter, so a store issued last in the loop body is what the next iteration's first wait would sit on
            if (o_prev) *o_prev = r_prev;
            r_prev = blend(c, v);
            o_prev = out_at(LIST ? s_cur : i);
            if constexpr (LIST) { s_cur = s_next; s_next = s_next2; }
        }
        if (o_prev) *o_prev = r_prev;
    }
}

// ---- fp32 backward: dtable[idx*F+f] += w * dout -----------------------------------------------------
// Generic-F fallback: one lane per (sample, level), F*8 independent atomics.
template <int F>
__global__ void __launch_bounds__(256) hash_bwd_f32_kernel(const float* __restrict__ xyzs, const float* __restrict__ dout,
                                                           ngp_hash_levels lv, int n, const int32_t* __restrict__ n_dev,
                                                           XyzNorm nm, float* __restrict__ dtable) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    if (n_dev) n = min(n, *n_dev);
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
        const float x = norm01(nm, xyzs[3 * (size_t)i]), y = norm01(nm, xyzs[3 * (size_t)i + 1]),
                    z = norm01(nm, xyzs[3 * (size_t)i + 2]);
        Corners c;
        corners<false>(L, level, lv.begin_fast_hash_level, x, y, z, c);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int f = 0; f < F; ++f) unsafeAtomicAdd(dtable + (size_t)c.idx[ci] * F + f, c.w[ci] * g[f]);
    }
}

// F = 2 fast path, shaped by what the MI355X atomic pipeline charges for (profiles/microbench/atomics2.hip):
// a float atomic instruction costs one request per DISTINCT 64-byte line it touches (~21 G lines/s chip-wide),
// adjacent lanes on one line are free, and duplicate addresses inside an instruction are NOT merged.
//   * lane quad = (sample, x-corner bit, feature): the four lanes of a quad hit (e, f0) (e, f1) (e', f0) (e', f1)
//     where e' is the x-neighbour entry -- adjacent (dense levels) or e^small-mask (xor hash) -- so the quad lands
//     on one 64-B line 7 times out of 8: ~4 line requests per (sample, level) instead of 16 scattered atomics.
//   * one wave = 16 consecutive samples x one level; consecutive samples of a ray sit in the same cell on the
//     coarse/mid levels, so equal-cell runs are summed with a segmented wave scan and only the last lane of a
//     run issues atomics (removes the in-instruction duplicates and ~60 % of all requests).
// Summation order differs from a serial loop: tolerance-checked against the oracle (float atomics are
// order-nondeterministic in the reference too).
__global__ void __launch_bounds__(256) hash_bwd_f32x2_kernel(const float* __restrict__ xyzs, const float* __restrict__ dout,
                                                             ngp_hash_levels lv, int n, const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ idx, XyzNorm nm, int enc_pairs,
                                                             float* __restrict__ dtable, int32_t* __restrict__ found_inf) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const size_t plane = (size_t)n;
    if (n_dev) n = min(n, *n_dev);
    const int nl = lv.n_levels, bfhl = lv.begin_fast_hash_level;
    const int lane = threadIdx.x & 63;
    const int s_in = lane >> 2, xb = (lane >> 1) & 1, f = lane & 1;
    const int n_tiles = (n + 15) >> 4;
    const int waves_per_block = blockDim.x >> 6;
    for (int tile = blockIdx.x * waves_per_block + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * waves_per_block) {
        const int i = tile * 16 + s_in;                  // position in dout (and in the live list, when there is one)
        const bool valid = i < n;
        float x = 0.f, y = 0.f, z = 0.f;
        if (valid) {
            const size_t src = idx ? (size_t)idx[i] : (size_t)i;
            x = norm01(nm, xyzs[3 * src]); y = norm01(nm, xyzs[3 * src + 1]); z = norm01(nm, xyzs[3 * src + 2]);
        }
        for (int level = 0; level < nl; ++level) {
            const size_t gi = enc_pairs ? ((size_t)(level < 8 ? level : 15 - level) * plane + i) * 4 + (level < 8 ? 0 : 2) + f
                                        : (size_t)i * (nl * 2) + level * 2 + f;
            const float g = valid ? dout[gi] : 0.0f;
            if (found_inf && !isfinite(g)) *found_inf = 1;          // GradScaler's inf/nan check, done where the data passes
            const float scale = L.scale[level];
            const uint32_t res = L.res[level], size = L.size[level], mode = L.mode[level];
            const float px = x * scale + 0.5f, py = y * scale + 0.5f, pz = z * scale + 0.5f;
            const uint32_t cx = f2u_sat(floorf(px)), cy = f2u_sat(floorf(py)), cz = f2u_sat(floorf(pz));
            const float fx = px - (float)cx, fy = py - (float)cy, fz = pz - (float)cz;
            // run structure: head = first sample of the tile or a different cell than the previous sample
            const uint32_t pcx = __shfl_up(cx, 4, 64), pcy = __shfl_up(cy, 4, 64), pcz = __shfl_up(cz, 4, 64);
            const int pvalid = __shfl_up((int)valid, 4, 64);
            bool head = (s_in == 0) || !valid || !pvalid || cx != pcx || cy != pcy || cz != pcz;
            const int nhead = __shfl_down((int)head, 4, 64);
            const bool tail = valid && ((s_in == 15) || nhead);
            const float wx = xb ? fx : 1.0f - fx;
            const uint32_t gx = cx + (uint32_t)xb;
            float v[4];
            uint32_t e[4];
            const bool dense = level < bfhl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {              // k = (z bit, y bit)
                const int yb = k & 1, zb = k >> 1;
                const float w = (1.0f * wx) * (yb ? fy : 1.0f - fy) * (zb ? fz : 1.0f - fz);   // same product order as fwd
                const uint32_t gy = cy + (uint32_t)yb, gz = cz + (uint32_t)zb;
                uint32_t h = dense ? (gx + gy * res + gz * res * res) : (gx ^ (gy * 2654435761u) ^ (gz * 805459861u));
                if (mode == 1u) h &= (size - 1u);
                else if (mode == 0u) { if (h >= size) { h -= size; if (h >= size) h %= size; } }
                else h = h % size;
                e[k] = L.offset[level] + h;
                v[k] = w * g;
            }
            // segmented inclusive scan over samples (lane distance 4 = one sample)
            bool hf = head;
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) {
                const int hup = __shfl_up((int)hf, d, 64);
                float vup[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) vup[k] = __shfl_up(v[k], d, 64);
                if (lane >= d && !hf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += vup[k];
                    hf = hup != 0;
                }
            }
            if (tail) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] != 0.0f) unsafeAtomicAdd(dtable + (size_t)e[k] * 2 + f, v[k]);
            }
        }
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
            // f32 product, THEN the cast to the table's f16 (two roundings, like the reference: see f16_bits_rn)
            const uint32_t bits = f16_bits_rn(c.w[ci] * g.x) | (f16_bits_rn(c.w[ci] * g.y) << 16);
            const half2v val = __builtin_bit_cast(half2v, bits);
            if (val.x == (_Float16)0 && val.y == (_Float16)0) continue;             // :212
            __builtin_amdgcn_global_atomic_fadd_v2f16(
                (__attribute__((address_space(1))) half2v*)(dtable + c.idx[ci]), val);   // global_atomic_pk_add_f16
        }
    }
}

// half2 backward, fused-path form: same arithmetic per contribution as hash_bwd_f16_kernel (g = f16(dout), val = f16(w * g),
// zero skip, packed f16x2 atomic), shaped like hash_bwd_f32x2_kernel: a lane pair = (sample, x-corner bit) -- the two entries are
// adjacent 4-byte words, one 64-byte line 7 times out of 8 -- a wave = 32 consecutive samples x one level, and equal-cell
// runs of consecutive samples are summed (in f32, rounded once) with a segmented wave scan before the single atomic.
// dout is the fp32 d_enc of the fused MLP backward (natural or pair-major layout).
__global__ void __launch_bounds__(256) hash_bwd_f16x2_kernel(const float* __restrict__ xyzs, const float* __restrict__ dout,
                                                             ngp_hash_levels lv, int n, const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ idx, XyzNorm nm, int enc_pairs,
                                                             __half2* __restrict__ dtable, int32_t* __restrict__ found_inf) {
    __shared__ LevelLDS L;
    load_levels(lv, L);
    const size_t plane = (size_t)n;
    if (n_dev) n = min(n, *n_dev);
    const int nl = lv.n_levels, bfhl = lv.begin_fast_hash_level;
    const int lane = threadIdx.x & 63;
    const int s_in = lane >> 1, xb = lane & 1;
    const int n_tiles = (n + 31) >> 5;
    const int waves_per_block = blockDim.x >> 6;
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    for (int tile = blockIdx.x * waves_per_block + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * waves_per_block) {
        const int i = tile * 32 + s_in;
        const bool valid = i < n;
        float x = 0.f, y = 0.f, z = 0.f;
        if (valid) {
            const size_t src = idx ? (size_t)idx[i] : (size_t)i;
            x = norm01(nm, xyzs[3 * src]); y = norm01(nm, xyzs[3 * src + 1]); z = norm01(nm, xyzs[3 * src + 2]);
        }
        for (int level = 0; level < nl; ++level) {
            float2 g = make_float2(0.f, 0.f);
            if (valid) {
                const float* gp = enc_pairs ? dout + ((size_t)(level < 8 ? level : 15 - level) * plane + i) * 4 + (level < 8 ? 0 : 2)
                                            : dout + (size_t)i * (nl * 2) + level * 2;
                g = *reinterpret_cast<const float2*>(gp);
            }
            g = __half22float2(__floats2half2_rn(g.x, g.y));              // the encoder's output gradient is an fp16 tensor
            if (found_inf && !(isfinite(g.x) && isfinite(g.y))) *found_inf = 1;
            const float scale = L.scale[level];
            const uint32_t res = L.res[level], size = L.size[level], mode = L.mode[level];
            const float px = x * scale + 0.5f, py = y * scale + 0.5f, pz = z * scale + 0.5f;
            const uint32_t cx = f2u_sat(floorf(px)), cy = f2u_sat(floorf(py)), cz = f2u_sat(floorf(pz));
            // hash_encoder_half.py:133: the cell is cast to f16 before the subtract
            const float fx = px - __half2float(__float2half_rn((float)cx)), fy = py - __half2float(__float2half_rn((float)cy)),
                        fz = pz - __half2float(__float2half_rn((float)cz));
            const uint32_t pcx = __shfl_up(cx, 2, 64), pcy = __shfl_up(cy, 2, 64), pcz = __shfl_up(cz, 2, 64);
            const int pvalid = __shfl_up((int)valid, 2, 64);
            const bool head = (s_in == 0) || !valid || !pvalid || cx != pcx || cy != pcy || cz != pcz;
            const int nhead = __shfl_down((int)head, 2, 64);
            const bool tail = valid && ((s_in == 31) || nhead);
            const float wx = xb ? fx : 1.0f - fx;
            const uint32_t gx = cx + (uint32_t)xb;
            float v0[4], v1[4];
            uint32_t e[4];
            const bool dense = level < bfhl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {              // k = (z bit, y bit)
                const int yb = k & 1, zb = k >> 1;
                const float w = (1.0f * wx) * (yb ? fy : 1.0f - fy) * (zb ? fz : 1.0f - fz);
                const uint32_t gy = cy + (uint32_t)yb, gz = cz + (uint32_t)zb;
                uint32_t h = dense ? (gx + gy * res + gz * res * res) : (gx ^ (gy * 2654435761u) ^ (gz * 805459861u));
                if (mode == 1u) h &= (size - 1u);
                else if (mode == 0u) { if (h >= size) { h -= size; if (h >= size) h %= size; } }
                else h = h % size;
                e[k] = L.offset[level] + h;
                const float2 r = make_float2(f16_round(w * g.x), f16_round(w * g.y));          // cast(w * g, f16) :205-208 (f32 product first)
                v0[k] = r.x; v1[k] = r.y;
            }
            bool hf = head;
#pragma unroll
            for (int d = 2; d < 64; d <<= 1) {
                const int hup = __shfl_up((int)hf, d, 64);
                float u0[4], u1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { u0[k] = __shfl_up(v0[k], d, 64); u1[k] = __shfl_up(v1[k], d, 64); }
                if (lane >= d && !hf) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k] += u0[k]; v1[k] += u1[k]; }
                    hf = hup != 0;
                }
            }
            if (tail) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    half2v val;
                    val.x = (_Float16)v0[k]; val.y = (_Float16)v1[k];
                    if (val.x == (_Float16)0 && val.y == (_Float16)0) continue;                 // :212
                    __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v*)(dtable + e[k]), val);
                }
            }
        }
    }
}

// any non-finite value in an f16 gradient buffer -> *found_inf = 1 (f16 sums overflow easily: GradScaler's check has to see
// the ACCUMULATED gradient, train.py:199)
__global__ void __launch_bounds__(256) check_finite_f16_kernel(const uint4* __restrict__ g, long n8, int32_t* __restrict__ found_inf) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = g[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) bad |= ((w[k] & 0x7c00u) == 0x7c00u) || ((w[k] & 0x7c000000u) == 0x7c000000u);   // exponent all ones
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1;
}

// The round-4 loop of hash_fwd_f32_xcd_kernel addresses the table with 32-bit byte offsets: tables of 4 GB and more (and
// NGP_EXPERIMENT hash_fwd_v1=1, for A/B timing) take the round-1..3 loop.
inline bool xcd_v1(const ngp_hash_levels& lv, unsigned entry_bytes) {
    static const bool forced = [] { const char* e = ngp_experiment("hash_fwd_v1"); return e && e[0] == '1'; }();
    return forced || (unsigned long long)(unsigned)lv.total_entries * entry_bytes >= 0xffffff00ull;
}

// tiles of 128 samples per level pair in one launch of hash_fwd_f32_xcd_kernel (8 workgroups per tile); NGP_EXPERIMENT hash_fwd_tiles for A/B runs.
// 768 since round 6 (512 before): three instead of two residency rounds of workgroups per CU at C2 -- gather 84.6-86.8 -> 79.8-84.2 us in
// four alternating pairs on one box (step 0.4696 -> 0.4631 ms); 1024 / 1536 no better; 65 536 rays and the C3 chunks unchanged.
inline int xcd_tiles_cap() {
    static const int cap = [] { const char* e = ngp_experiment("hash_fwd_tiles"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
    return cap;
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

int ngp_hash_fwd_f32_ex(const float* xyzs, const float* table, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                        int normalize, float lo, float hi, int enc_pairs, float* out, void* stream) {
    if (n_max <= 0) return 0;
    if (lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n_max * lv->n_levels, 256);
    hipStream_t s = (hipStream_t)stream;
    XyzNorm nm = {normalize, lo, hi};
#ifdef NGP_HASH_FWD_DIAG
    if (const char* e = ngp_experiment("hash_fwd_free_levels")) nm.enabled |= (int)(strtoul(e, nullptr, 0) & 0xffffu) << 8;
#endif
    if (enc_pairs && !(lv->n_features == 2 && lv->n_levels == 16)) return -1;
    if (lv->n_features == 2 && lv->n_levels == 16 && (n_max >= 4096 || enc_pairs)) {
        int tiles = (n_max + 127) / 128;
        if (tiles > xcd_tiles_cap()) tiles = xcd_tiles_cap();   // 8 x cap blocks, tile-stride beyond
        if (xcd_v1(*lv, 8)) hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<0, true>), dim3(8 * tiles), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, enc_pairs, out);
        else hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<0, false>), dim3(8 * tiles), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, enc_pairs, out);
        NGP_LAUNCH_CHECK();
        return 0;
    }
    switch (lv->n_features) {
        case 1: hipLaunchKernelGGL(hash_fwd_f32_kernel<1>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, out); break;
        case 2: hipLaunchKernelGGL(hash_fwd_f32_kernel<2>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, out); break;
        case 4: hipLaunchKernelGGL(hash_fwd_f32_kernel<4>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, out); break;
        case 8: hipLaunchKernelGGL(hash_fwd_f32_kernel<8>, dim3(grid), dim3(256), 0, s, xyzs, table, *lv, n_max, n_dev, nm, out); break;
        default: return -1;
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

// The fused path's encoder over a LIST of samples (round 5): rows list[0 .. *n_list) of xyzs are encoded into the same rows of out
// (16 levels x 2 features, natural or pair-major planes of stride n_max).  table_kind 0: fp32 table, 1: its bf16 storage copy.
// Returns -2 where the specialised kernel does not apply (other table shapes, tables of 4 GB and more): the caller encodes everything.
int ngp_hash_fwd_list(const float* xyzs, const void* table, int table_kind, const ngp_hash_levels* lv, int n_max, const int32_t* n_list,
                      const int32_t* list, int normalize, float lo, float hi, int enc_pairs, float* out, void* stream) {
    if (n_max <= 0) return 0;
    if (!n_list || !list || table_kind < 0 || table_kind > 1) return -1;
    if (!(lv->n_features == 2 && lv->n_levels == 16) || xcd_v1(*lv, table_kind ? 4 : 8)) return -2;
    const XyzNorm nm = {normalize, lo, hi};
    int tiles = (n_max + 127) / 128;
    if (tiles > xcd_tiles_cap()) tiles = xcd_tiles_cap();
    const float* t = reinterpret_cast<const float*>(table);
    if (table_kind) hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<1, false, true>), dim3(8 * tiles), dim3(256), 0, (hipStream_t)stream, xyzs, t, *lv, n_max, n_list, nm, enc_pairs, out, list);
    else hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<0, false, true>), dim3(8 * tiles), dim3(256), 0, (hipStream_t)stream, xyzs, t, *lv, n_max, n_list, nm, enc_pairs, out, list);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_fwd_bf16_ex(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                         int normalize, float lo, float hi, int enc_pairs, float* out, void* stream) {
    if (n_max <= 0) return 0;
    if (lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS || lv->n_features != 2) return -1;
    hipStream_t s = (hipStream_t)stream;
    const XyzNorm nm = {normalize, lo, hi};
    const float* t = reinterpret_cast<const float*>(table);
    if (enc_pairs && lv->n_levels != 16) return -1;
    if (lv->n_levels == 16 && (n_max >= 4096 || enc_pairs)) {
        int tiles = (n_max + 127) / 128;
        if (tiles > xcd_tiles_cap()) tiles = xcd_tiles_cap();
        if (xcd_v1(*lv, 4)) hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<1, true>), dim3(8 * tiles), dim3(256), 0, s, xyzs, t, *lv, n_max, n_dev, nm, enc_pairs, out);
        else hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<1, false>), dim3(8 * tiles), dim3(256), 0, s, xyzs, t, *lv, n_max, n_dev, nm, enc_pairs, out);
    } else {
        const int grid = grid_for((long long)n_max * lv->n_levels, 256);
        hipLaunchKernelGGL((hash_fwd_f32_kernel<2, true>), dim3(grid), dim3(256), 0, s, xyzs, t, *lv, n_max, n_dev, nm, out);
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_fwd_f32(const float* xyzs, const float* table, const ngp_hash_levels* lv, int n, float* out, void* stream) {
    return ngp_hash_fwd_f32_ex(xyzs, table, lv, n, nullptr, 0, 0.0f, 1.0f, 0, out, stream);
}

int ngp_hash_bwd_f32_live(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                          const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, float* dtable, int32_t* found_inf,
                          void* stream) {
    if (n_max <= 0) return 0;
    if (lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    const int grid = grid_for((long long)n_max * lv->n_levels, 256);
    hipStream_t s = (hipStream_t)stream;
    const XyzNorm nm = {normalize, lo, hi};
    if (enc_pairs && !(lv->n_features == 2 && lv->n_levels == 16)) return -1;
    if (live_idx && lv->n_features != 2) return -1;
    switch (lv->n_features) {
        case 1: hipLaunchKernelGGL(hash_bwd_f32_kernel<1>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n_max, n_dev, nm, dtable); break;
        case 2: {
            const int tiles = (n_max + 15) / 16;
            const int g2 = tiles < 4 ? 1 : (tiles / 4 < 8192 ? (tiles + 3) / 4 : 8192);
            hipLaunchKernelGGL(hash_bwd_f32x2_kernel, dim3(g2), dim3(256), 0, s, xyzs, dout, *lv, n_max, n_dev, live_idx, nm, enc_pairs, dtable, found_inf);
            break;
        }
        case 4: hipLaunchKernelGGL(hash_bwd_f32_kernel<4>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n_max, n_dev, nm, dtable); break;
        case 8: hipLaunchKernelGGL(hash_bwd_f32_kernel<8>, dim3(grid), dim3(256), 0, s, xyzs, dout, *lv, n_max, n_dev, nm, dtable); break;
        default: return -1;
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_f32_ex(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                        int normalize, float lo, float hi, int enc_pairs, float* dtable, int32_t* found_inf, void* stream) {
    return ngp_hash_bwd_f32_live(xyzs, dout, lv, n_max, n_dev, nullptr, normalize, lo, hi, enc_pairs, dtable, found_inf, stream);
}

int ngp_hash_bwd_f32(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n, float* dtable, void* stream) {
    return ngp_hash_bwd_f32_ex(xyzs, dout, lv, n, nullptr, 0, 0.0f, 1.0f, 0, dtable, nullptr, stream);
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

int ngp_hash_fwd_f16_ex(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                        int normalize, float lo, float hi, int enc_pairs, float* out, void* stream) {
    if (n_max <= 0) return 0;
    if (lv->n_levels != 16 || lv->n_features != 2) return -1;
    int tiles = (n_max + 127) / 128;
    if (tiles > xcd_tiles_cap()) tiles = xcd_tiles_cap();
    const XyzNorm nm = {normalize, lo, hi};
    if (xcd_v1(*lv, 4))
        hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<2, true>), dim3(8 * tiles), dim3(256), 0, (hipStream_t)stream, xyzs,
                           reinterpret_cast<const float*>(table), *lv, n_max, n_dev, nm, enc_pairs, out);
    else
        hipLaunchKernelGGL((hash_fwd_f32_xcd_kernel<2, false>), dim3(8 * tiles), dim3(256), 0, (hipStream_t)stream, xyzs,
                           reinterpret_cast<const float*>(table), *lv, n_max, n_dev, nm, enc_pairs, out);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_f16_live(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                          const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, uint16_t* dtable, int32_t* found_inf,
                          void* stream) {
    if (n_max <= 0) return 0;
    if (lv->n_features != 2 || lv->n_levels < 1 || lv->n_levels > NGP_MAX_LEVELS) return -1;
    if (enc_pairs && lv->n_levels != 16) return -1;
    const XyzNorm nm = {normalize, lo, hi};
    const int tiles = (n_max + 31) / 32;
    const int g2 = tiles < 4 ? 1 : (tiles / 4 < 8192 ? (tiles + 3) / 4 : 8192);
    hipLaunchKernelGGL(hash_bwd_f16x2_kernel, dim3(g2), dim3(256), 0, (hipStream_t)stream, xyzs, dout, *lv, n_max, n_dev, live_idx, nm,
                       enc_pairs, (__half2*)dtable, found_inf);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_f16_ex(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int normalize,
                        float lo, float hi, int enc_pairs, uint16_t* dtable, int32_t* found_inf, void* stream) {
    return ngp_hash_bwd_f16_live(xyzs, dout, lv, n_max, n_dev, nullptr, normalize, lo, hi, enc_pairs, dtable, found_inf, stream);
}

int ngp_check_finite_f16(const uint16_t* g, long long n, int32_t* found_inf, void* stream) {
    if (n <= 0) return 0;
    if (n % 8 != 0) return -1;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(check_finite_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, (long)(n / 8),
                       found_inf);
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
