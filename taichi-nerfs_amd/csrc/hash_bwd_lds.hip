// hash_bwd_lds.hip -- hash-grid scatter-add (backward of the fp32 encoder, F = 2) WITHOUT global float atomics.
//
// Replaces the Taichi-autodiff backward of modules/hash_encoder.py:89-143 (call site :269): dtable[idx*2+f] += w * dout.
//
// Why a second formulation.  Round 1's kernel (hash_bwd_f32x2_kernel, hash_grid.hip) issues one global float atomic per
// (sample, level, corner pair) and sits at the chip's atomic line-request rate (~21 G 64-byte lines/s, measured in
// profiles/microbench/atomics*.hip; it executes memory-side, no L2 locality helps): 1.7 ns per live sample, 590-650 us at
// the 350-400 k live samples of a real training step = 61 % of the step.  But the whole gradient table is only 45.7 MB and
// the chip has 256 x 160 KB = 40 MB of LDS.  So the table is cut into SLICES of 8192 entries, one workgroup OWNS one slice
// in its LDS, looks at the live samples of its level, accumulates the contributions that land in its slice with LDS
// atomics, and adds the slice to the table once, coalesced and non-atomically.
//
// What the hardware dictated (profiles/microbench/lds_atomics.hip -> profiles/r02_microbench_lds_atomics.txt):
//   * ds_add_f32 is processed ONE LANE AT A TIME (~3 clocks per active lane: 80 ns per full wave instruction per CU, 204 G
//     lane-adds/s chip-wide) -- the first version of this kernel, with f32 accumulators, took 950 us.  ds_add_f64 retires a
//     conflict-free wave instruction in ~25 clocks (8x faster; ds_add_u64 in ~13).  Hence DOUBLE accumulators: every
//     contribution w * g is formed in f32 exactly like the reference's product, widened (exact) and summed in f64; the slice is
//     rounded to f32 once when it is flushed.  The result no longer depends on the order of the adds beyond ~1e-16 relative
//     (float atomics differ run to run at ~1e-7) and is more accurate than an f32 running sum.
//   * same-address LDS atomics still serialise (~3 clocks per lane), so on the coarse levels, where consecutive samples of a
//     ray sit in one cell for many steps, equal-cell runs are pre-summed with a segmented DPP scan inside 16-lane rows.
//   * the kernel is VALU-bound once the atomics are cheap (PMC: 160 M VALU wave-instructions in the first version), so the
//     hashed levels use the structure of the hash: the slice of a corner pair depends on (y, z) only (x flips bits below the
//     slice bits), two multiplies serve all four (y, z) combinations, and a lane only runs the body of the combinations that
//     actually land in this slice.
//   * every slice owner has to look at every live sample of its level (64 owners per hashed level), so the filter must be
//     nearly free: the prepass writes one hit BIT per (level, slice, sample); an owner scans 4096 samples per 8-byte vector
//     load (one super-chunk ahead), compacts the hits into a per-wave LDS queue and processes them 128 at a time with all
//     lanes busy, the gathers of the next batch in flight while the current one is accumulated.
//
//   prep  : compact normalised positions xyzc[i]; per (level, slice) hit bitmaps
//   main  : task = (level, slice, replica r of R); looks at samples [r S/R, (r+1) S/R); coarse levels whose few slices would
//           see every sample are replicated over sample ranges (private LDS copy each, flushed with float atomics: a few
//           thousand coalesced lines per replica).
// Task order: blockIdx b lands on XCD b % 8; all slices of one level go to one XCD where possible so that its owners stream
// the same position / gradient lines out of that XCD's L2.
//
// Round 3, measured and NOT shipped (commit 3017399 has the code; profiles/r03_scatter_add_experiments.txt the numbers): hit LISTS
// instead of bitmaps for the hashed levels (a counting sort of every 2048-sample chunk by slice in the prepass, an owner reads its
// segment of every chunk and does one corner pair per entry: no scan, no combination test, 70 M instead of ~100 M VALU
// instructions per launch), with a three-generation software pipeline (8 gathers per lane in flight) and a feature-blocked LDS
// image.  Correct, prepass +2.5 us -- and the main launch 6 % SLOWER on identical inputs (231-237 vs 218-223 us).  The -DNGP_BWD_DIAG
// breakdown says why: with the accumulate switched off but the gathers still issued both forms take 177 us, with the gathers off
// too 123 (bitmaps) / 91 us (lists).  The launch is bound by its two 64-line gathers per 64 hits (48 M vector-L1 accesses, 27 M L2
// requests per launch = ~27 TB/s of the L2's ~34 TB/s while the hashed levels run), not by instruction issue, LDS banking or
// memory-level parallelism: LDS, vector L1 and VALU are each 40-50 % busy and the waves sit in s_waitcnt 55 % of their cycles.
// A second form (commit 2df8868) gave the list owners 4096-entry slices so that TWO of them (46 VGPRs, 8 waves per SIMD) share a
// CU: 58 % slower (333 vs 210 us) -- a task of half the hits takes as long as a task did before.  And a task takes the same time
// whether 32 or 256 CUs are working.  So the bound is a per-CU THROUGHPUT that neither occupancy nor instruction count moves:
// ~290 cycles per 64 hits and CU, which is what 128 scattered vector-L1 lane accesses (two gathers, ~55 % missing the L1, each miss
// a 128-byte fill for 8-12 useful bytes) + 100 LDS cycles of f64 atomics + ~110 VALU cycles add up to when the three serialise
// inside each wave's gather -> weights -> adds chain.
#include "ngp_device.h"
#include "hash_common.h"
#include <stdlib.h>
#include <string.h>

namespace ngp {

constexpr int BW_SLICE_LOG2 = 13;
constexpr int BW_SLICE_ENTRIES = 1 << BW_SLICE_LOG2;          // 8192 entries x 2 features x f64 = 128 KB of LDS
constexpr int BW_MAX_SLICES = 64;                              // per level: 2^19 entries
constexpr int BW_THREADS = 1024;
constexpr int BW_WAVES = BW_THREADS / 64;
constexpr int BW_Q = 256;                                      // per-wave hit queue (entries)
constexpr int BW_MAX_TASKS = 1536;
constexpr int BW_PREP_BLOCKS = 2048;
constexpr size_t BW_CTR_BYTES = 64;                           // 8 queue heads + the exit counter of the main kernel + [9]: a flush-Adam slice sum overflowed
constexpr int BW_PERSISTENT_BLOCKS = 256;                     // one 1024-thread workgroup (147 KB of LDS) per CU

struct BwdPlan {
    int32_t n_blocks;
    uint32_t merge_mask;                  // bit l: pre-sum equal-cell runs on level l
    uint32_t diag;                        // timing experiments (-DNGP_BWD_DIAG builds): 1 no LDS adds, 2 no gathers, 4 no accumulate
    uint32_t det;                         // ngp_hash_bwd_sliced_deterministic: run pre-summing groups hits per super-chunk (see bwd_task)
    int32_t merge_chunks;                 // see LevelParams
    uint8_t nrep[NGP_MAX_LEVELS];         // replicas (sample ranges) per slice of level l
    uint16_t task[BW_MAX_TASKS];          // level | slice << 4 | rep << 10; XCD x owns task[xoff[x] .. xoff[x] + xlen[x])
    uint16_t xoff[8], xlen[8];
};

__device__ __forceinline__ uint32_t level_index(bool dense, uint32_t mode, uint32_t size, uint32_t res, uint32_t gx, uint32_t gy,
                                                uint32_t gz) {
    uint32_t h = dense ? (gx + gy * res + gz * res * res) : (gx ^ (gy * 2654435761u) ^ (gz * 805459861u));   // :53-60 / :43-51
    if (mode == 1u) h &= (size - 1u);
    else if (mode == 0u) { if (h >= size) { h -= size; if (h >= size) h %= size; } }
    else h = h % size;                                                                                       // :71
    return h;
}

// dense level whose table holds the whole grid (mode 0: size >= res^3, so x + y res + z res^2 < 2 size and `% size` is one
// conditional subtract): the eight corner indices from one base index, branch-free
__device__ __forceinline__ uint32_t dense0_index(uint32_t base, uint32_t res, uint32_t res2, uint32_t size, int c) {
    uint32_t h = base + (uint32_t)(c & 1) + (((c >> 1) & 1) ? res : 0u) + ((c >> 2) ? res2 : 0u);
    return h >= size ? h - size : h;
}

// Which slice owns entry h of a level, and where in the owner's LDS image it lives.
//   hashed levels : contiguous ranges of 8192 entries (slice = h >> 13): the xor hash already spreads space evenly over them, and
//                   both x corners of a (y, z) combination fall into the same range;
//   dense levels  : a contiguous range is a z-slab of the grid, and a scene that sits in part of the z range loads a few owners
//                   with most of the samples (measured: 200 us stragglers on level 5).  Where a level has many slices, blocks
//                   of about one z-plane (a power of two of entries) are dealt round-robin to the ns owners instead.
struct SliceMap {
    uint32_t ns, magic;      // magic = ceil(2^32 / ns): blk / ns = umulhi(blk, magic) for blk < 2^32 / ns^2
    uint32_t bshift;         // log2 of the interleaving block (entries)
    bool interleaved;
};
__device__ __forceinline__ uint32_t slice_of(const SliceMap M, uint32_t h, uint32_t& local) {
    if (!M.interleaved) { local = h & (BW_SLICE_ENTRIES - 1); return h >> BW_SLICE_LOG2; }
    const uint32_t blk = h >> M.bshift, qd = __umulhi(blk, M.magic), r = blk - qd * M.ns;
    local = (qd << M.bshift) | (h & ((1u << M.bshift) - 1u));
    return r;
}
__device__ __forceinline__ uint32_t entry_of(const SliceMap M, uint32_t sl, uint32_t local) {      // inverse of slice_of
    if (!M.interleaved) return sl * (uint32_t)BW_SLICE_ENTRIES + local;
    return (((local >> M.bshift) * M.ns + sl) << M.bshift) | (local & ((1u << M.bshift) - 1u));
}
__host__ __device__ __forceinline__ SliceMap slice_map(uint32_t size, uint32_t res, bool dense) {
    SliceMap M;
    M.ns = (size + BW_SLICE_ENTRIES - 1) >> BW_SLICE_LOG2;
    // interleave only where there are many slices (and therefore few sample-range replicas to even out the load): a sample of
    // a dense level touches the z and z + 1 planes, so with blocks of about one z-plane it still falls into 1-2 slices
    M.interleaved = dense && M.ns >= 8;
    const uint32_t plane = res * res, pl = plane > 128u ? plane : 128u, lg = 31u - (uint32_t)__builtin_clz(pl);
    M.bshift = lg < (uint32_t)BW_SLICE_LOG2 ? lg : (uint32_t)BW_SLICE_LOG2;
    M.magic = M.ns > 1 ? (uint32_t)((0x100000000ull + M.ns - 1) / M.ns) : 0u;
    return M;
}

// The prepass's per-level constants, made once on the host and handed over by value (kernel arguments are read with scalar loads:
// everything here lands in SGPRs).  Round 1-3 the kernel rebuilt them per level per 64-sample tile from the level table in LDS;
// slice_map's 64-bit division alone was ~110 dependent scalar instructions of every level trip (70 % of the launch's 13 M SALU
// instructions, profiles/r04_pmc.json).
struct PrepLevel {
    float scale;
    uint32_t res, size, mode;        // mode as load_levels derives it: 0 dense (conditional subtract), 1 hashed pow2 (mask), 2 real modulo
    SliceMap map;
};
struct PrepLevels { PrepLevel l[NGP_MAX_LEVELS]; };
static PrepLevels make_prep_levels(const ngp_hash_levels& lv) {
    PrepLevels P = {};
    for (int t = 0; t < lv.n_levels && t < NGP_MAX_LEVELS; ++t) {
        PrepLevel& q = P.l[t];
        q.scale = lv.scale[t]; q.res = lv.resolution[t]; q.size = lv.map_size[t];
        const bool dense = t < lv.begin_fast_hash_level;
        if (dense) { const uint64_t r = q.res; q.mode = ((uint64_t)q.size >= r * r * r && r >= 2) ? 0u : 2u; }
        else q.mode = (q.size != 0 && (q.size & (q.size - 1)) == 0) ? 1u : 2u;
        q.map = slice_map(q.size, q.res, dense);
    }
    return P;
}

__device__ __forceinline__ const float* grad_ptr(const float* dout, int level, size_t i, size_t plane, int enc_pairs, int nl) {
    return enc_pairs ? dout + ((size_t)(level < 8 ? level : 15 - level) * plane + i) * 4 + (level < 8 ? 0 : 2)
                     : dout + i * (size_t)(nl * 2) + level * 2;
}

// ---- prepass: compact positions, per-(level, slice) hit bitmaps -------------------------------------------------------------
// bitmap[(level * 64 + slice) * wstride + t] bit j = sample 64 t + j has a corner in that slice.  A wave owns 64 consecutive
// samples.  Levels of many slices: each lane ORs its corners' bits into a 64-word LDS table (one word per slice; random
// slices, few same-word collisions), lane s then stores word s.  Levels of <= 8 slices (where every lane would hit the same
// few words and the LDS atomics serialise): one wave ballot per slice.
// (Round 4, measured and not kept: FOUR tiles per wave trip with lane s storing the four words of slice s as one 32-byte piece --
// 6.4 M scattered 8-byte stores become 1.6 M -- is correct and SLOWER, 65 vs 50 us at 433 k samples: the launch then has 1700
// working waves instead of 6800, and a trip is a serial chain over 16 levels.  The prepass is latency-, not store-bound.)
// Levels are dealt to the ballot form (<= 8 slices) or the LDS form (> 8) by their slice count, which grows with the level: lds_begin
// is the first LDS-form level (the host checks that the split is a prefix / suffix).  The LDS-form levels are done BATCH at a time:
// all their ORs, one fence, then every level's row is read, stored and cleared.  BATCH = 1 is rounds 2-3's order and the default:
// measured at 750 k samples (profiles/r04_hash_fwd_loop_experiment.txt) BATCH 1 / 3 / 6 / 12 = 59.2 / 59.7 / 63.2 / 66.7 us -- the
// per-level chain (clear, fence, ORs, fence, read, store) is NOT what the launch waits for; its LDS atomics and its 8-byte
// scattered row stores are (larger batches only cost occupancy: 24 KB of LDS per workgroup at 12).
template <int BATCH>
__global__ void __launch_bounds__(256) hash_bwd_prep_kernel(const float* __restrict__ xyzs, const int32_t* __restrict__ idx,
                                                            PrepLevels pl, int nl, int bfhl, int lds_begin, int n,
                                                            const int32_t* __restrict__ n_dev, XyzNorm nm, size_t wstride,
                                                            uint32_t single_slice_levels, float* __restrict__ xyzc,
                                                            unsigned long long* __restrict__ bitmap, uint32_t* __restrict__ ctr) {
    __shared__ unsigned long long words[4][BATCH][BW_MAX_SLICES];
    if (blockIdx.x == 0 && threadIdx.x < 16) ctr[threadIdx.x] = 0u;       // the main kernel's queue heads (it also resets them itself)
    if (n_dev) n = min(n, *n_dev);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = (n + 63) >> 6;
#pragma unroll
    for (int k = 0; k < BATCH; ++k) words[wave][k][lane] = 0ull;         // (a wave only ever touches its own rows)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        const int i = tile * 64 + lane;
        const bool valid = i < n;
        float x = 0.f, y = 0.f, z = 0.f;
        if (valid) {
            const size_t src = idx ? (size_t)idx[i] : (size_t)i;
            const float rx = xyzs[3 * src], ry = xyzs[3 * src + 1], rz = xyzs[3 * src + 2];   // one request, then norm01's branches
            x = norm01(nm, rx); y = norm01(nm, ry); z = norm01(nm, rz);
            xyzc[3 * (size_t)i] = x; xyzc[3 * (size_t)i + 1] = y; xyzc[3 * (size_t)i + 2] = z;
        }
        // ---- levels of <= 8 slices: one wave ballot per slice
        for (int level = 0; level < lds_begin; ++level) {
            if ((single_slice_levels >> level) & 1u) continue;             // every sample is a hit there: no bitmap needed
            const PrepLevel& P = pl.l[level];
            const uint32_t res = P.res, size = P.size, mode = P.mode;
            const uint32_t cx = f2u_sat(floorf(x * P.scale + 0.5f)), cy = f2u_sat(floorf(y * P.scale + 0.5f)),
                           cz = f2u_sat(floorf(z * P.scale + 0.5f));
            const bool dense = level < bfhl;
            const SliceMap SM = P.map;
            unsigned long long* row = bitmap + ((size_t)level * BW_MAX_SLICES) * wstride + tile;
            const uint32_t res2 = res * res, base = cx + cy * res + cz * res2;
            uint32_t m = 0u;
            if (valid) {
                if (dense && mode == 0u) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) { uint32_t loc; m |= 1u << slice_of(SM, dense0_index(base, res, res2, size, c), loc); }
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        uint32_t loc;
                        m |= 1u << slice_of(SM, level_index(dense, mode, size, res, cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2)), loc);
                    }
                }
            }
            for (int s = 0; s < (int)SM.ns; ++s) {
                const unsigned long long w = __ballot((m >> s) & 1u);
                if (lane == 0) row[(size_t)s * wstride] = w;
            }
        }
        // ---- levels of many slices: each lane ORs its corners' bits into the level's 64-word LDS table
        for (int lb = lds_begin; lb < nl; lb += BATCH) {
            const int le = min(lb + BATCH, nl);
            if (valid) {
                for (int level = lb; level < le; ++level) {
                    const PrepLevel& P = pl.l[level];
                    const uint32_t res = P.res, size = P.size, mode = P.mode;
                    const uint32_t cx = f2u_sat(floorf(x * P.scale + 0.5f)), cy = f2u_sat(floorf(y * P.scale + 0.5f)),
                                   cz = f2u_sat(floorf(z * P.scale + 0.5f));
                    const bool dense = level < bfhl;
                    const SliceMap SM = P.map;
                    unsigned long long* wd = words[wave][level - lb];
                    if (!dense && mode == 1u && res < (1u << BW_SLICE_LOG2)) {
                        // xor hash, power-of-two table: x only flips bits below the slice bits -> one slice per (y, z) combination
                        const uint32_t b0 = cy * 2654435761u, b1 = b0 + 2654435761u, c0 = cz * 805459861u, c1 = c0 + 805459861u;
                        const uint32_t msk = size - 1u;
                        atomicOr(&wd[((b0 ^ c0) & msk) >> BW_SLICE_LOG2], 1ull << lane);         // ds_or_b64
                        atomicOr(&wd[((b1 ^ c0) & msk) >> BW_SLICE_LOG2], 1ull << lane);
                        atomicOr(&wd[((b0 ^ c1) & msk) >> BW_SLICE_LOG2], 1ull << lane);
                        atomicOr(&wd[((b1 ^ c1) & msk) >> BW_SLICE_LOG2], 1ull << lane);
                    } else if (dense && mode == 0u) {
                        // the x pair is adjacent and stays inside one interleaving block except at a block edge: 4 ORs, + the edge cases
                        const uint32_t res2 = res * res, base = cx + cy * res + cz * res2;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t loc;
                            const uint32_t s0 = slice_of(SM, dense0_index(base, res, res2, size, 2 * k), loc);
                            const uint32_t s1 = slice_of(SM, dense0_index(base, res, res2, size, 2 * k + 1), loc);
                            atomicOr(&wd[s0], 1ull << lane);
                            if (s1 != s0) atomicOr(&wd[s1], 1ull << lane);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            uint32_t loc;
                            const uint32_t h = level_index(dense, mode, size, res, cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2));
                            atomicOr(&wd[slice_of(SM, h, loc)], 1ull << lane);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int level = lb; level < le; ++level) {
                const unsigned long long w = words[wave][level - lb][lane];
                words[wave][level - lb][lane] = 0ull;                              // clean for the next batch / tile
                if (lane < (int)pl.l[level].map.ns)
                    bitmap[((size_t)level * BW_MAX_SLICES + lane) * wstride + tile] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- main kernel -------------------------------------------------------------------------------------------------------
struct LevelParams {
    float scale;
    uint32_t res, size, mode, offset;
    bool dense;
    SliceMap map;
    uint32_t diag;           // diagnostics only (NGP_BWD_DIAG): bit 0 = skip the LDS adds, bit 1 = skip the gathers
    uint32_t det;
    int merge_chunks;        // run-pre-summing levels: at least this many super-chunks per task (0: the slice-count rule alone)
};
struct Hit {
    float x, y, z, g0, g1;
};
struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
};

__device__ __forceinline__ float round16(float v) { return f16_round(v); }      // through fp16 (RNE) and back, never fused into the product

__device__ __forceinline__ Hit load_hit(const int level, const int i, const bool valid, const float* __restrict__ xyzc,
                                        const float* __restrict__ dout, const size_t plane, const int enc_pairs, const int nl,
                                        int32_t* __restrict__ found_inf, const uint32_t diag = 0u, const bool half = false) {
    Hit h = {0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef NGP_BWD_DIAG
    if (diag & 2u) { const float t = (float)(i & 1023) * (1.0f / 1024.0f); h.x = t; h.y = 1.0f - t; h.z = 0.5f * t; h.g0 = 1.0f; h.g1 = t; return h; }
#endif
    if (valid) {
        const F3 p = *reinterpret_cast<const F3*>(xyzc + 3 * (size_t)i);                  // one 12-byte gather
        h.x = p.x; h.y = p.y; h.z = p.z;
#ifdef NGP_BWD_DIAG
        if (diag & 8u) { h.g0 = 1.0f; h.g1 = 0.5f; return h; }      // timing experiment: ONE gather per hit (no gradient load)
#endif
        const float2 g = *reinterpret_cast<const float2*>(grad_ptr(dout, level, (size_t)i, plane, enc_pairs, nl));
        h.g0 = g.x; h.g1 = g.y;
        if (half) { h.g0 = round16(h.g0); h.g1 = round16(h.g1); }      // half2 encoder: its output gradient is an fp16 tensor
    }
    return h;
}

__device__ __forceinline__ void lds_add(double* p, float v) { atomicAdd(p, (double)v); }          // ds_add_f64
#ifdef NGP_BWD_DIAG          // timing experiments only (profiles/microbench): lets the LDS adds / the gathers be switched off at run time
#define LDS_ADD(p, v) do { if (!(P.diag & 1u)) lds_add((p), (v)); else asm volatile("" :: "v"(v), "v"(p)); } while (0)
#else
#define LDS_ADD(p, v) lds_add((p), (v))
#endif

// DPP row_shr:D inside each 16-lane row; lanes without a source get `fill`
template <int D>
__device__ __forceinline__ float row_shr_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true));
}
template <int D>
__device__ __forceinline__ int row_shr_i(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x110 + D, 0xf, 0xf, false);
}

template <int D>
__device__ __forceinline__ void seg_step(float (&v0)[8], float (&v1)[8], int& hf) {
    const int hup = row_shr_i<D>(hf, 1);
    float u0[8], u1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { u0[c] = row_shr_f<D>(v0[c]); u1[c] = row_shr_f<D>(v1[c]); }
    if (!hf) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { v0[c] += u0[c]; v1[c] += u1[c]; }
        hf = hup;
    }
}

enum { KIND_GENERIC = 0, KIND_HASHED = 1, KIND_MERGE = 2, KIND_MERGE0 = 3,      // MERGE0: run pre-summing on a mode-0 dense level
       KIND_HALF = 4 };   // | KIND_HALF: the half2 encoder's arithmetic (hash_encoder_half.py:133,205-208): cell cast to f16 before
                          // the subtract, every w * g product rounded to f16; the owner's f64 sum is rounded to f16 once, at the flush

// One batch of <= 64 hits (one per lane): accumulate this level's contributions that fall into slice `sl`.
template <int KIND>
__device__ __forceinline__ void accumulate(const LevelParams P, const uint32_t sl, const bool single, const Hit H, const bool valid,
                                           double* __restrict__ slice) {
    constexpr bool HALF = (KIND & KIND_HALF) != 0;
    constexpr int K = KIND & 3;
    const int lane = threadIdx.x & 63;
    float g0 = H.g0, g1 = H.g1;
    const float px = H.x * P.scale + 0.5f, py = H.y * P.scale + 0.5f, pz = H.z * P.scale + 0.5f;
    uint32_t cx = f2u_sat(floorf(px)), cy = f2u_sat(floorf(py)), cz = f2u_sat(floorf(pz));
    const float fx = px - (HALF ? round16((float)cx) : (float)cx), fy = py - (HALF ? round16((float)cy) : (float)cy),
                fz = pz - (HALF ? round16((float)cz) : (float)cz);
    auto R = [](float v) { return HALF ? round16(v) : v; };
    const bool act = valid && (g0 != 0.0f || g1 != 0.0f);           // exact-zero gradients contribute nothing
    if (K == KIND_HASHED) {
        // xor hash into a power-of-two table with res < 2^13: h = (gx ^ A) & mask, A = gy P1 ^ gz P2; gx < 2^13 cannot reach the
        // slice bits, so both x corners of a (y, z) combination share the slice, and two multiplies serve all four combinations
        const uint32_t msk = P.size - 1u;
        const uint32_t b0 = cy * 2654435761u, b1 = b0 + 2654435761u, c0 = cz * 805459861u, c1 = c0 + 805459861u;
        const uint32_t A0 = b0 ^ c0, A1 = b1 ^ c0, A2 = b0 ^ c1, A3 = b1 ^ c1;             // k = (z bit, y bit)
        uint32_t m = 0u;
        if (act) {
            m = (uint32_t)(((A0 & msk) >> BW_SLICE_LOG2) == sl) | ((uint32_t)(((A1 & msk) >> BW_SLICE_LOG2) == sl) << 1) |
                ((uint32_t)(((A2 & msk) >> BW_SLICE_LOG2) == sl) << 2) | ((uint32_t)(((A3 & msk) >> BW_SLICE_LOG2) == sl) << 3);
        }
        const float wx0 = 1.0f * (1.0f - fx), wx1 = 1.0f * fx;                               // same product order as the forward
        while (__any(m != 0u)) {
            if (m != 0u) {
                const int k = __builtin_ctz(m);
                m &= m - 1u;
                const uint32_t A = (k & 2) ? ((k & 1) ? A3 : A2) : ((k & 1) ? A1 : A0);
                const float wy = (k & 1) ? fy : 1.0f - fy, wz = (k & 2) ? fz : 1.0f - fz;
                const float w0 = (wx0 * wy) * wz, w1 = (wx1 * wy) * wz;
                double* p0 = slice + 2 * (((cx ^ A) & msk) & (BW_SLICE_ENTRIES - 1));
                double* p1 = slice + 2 * ((((cx + 1u) ^ A) & msk) & (BW_SLICE_ENTRIES - 1));
                LDS_ADD(p0, R(w0 * g0)); LDS_ADD(p0 + 1, R(w0 * g1));
                LDS_ADD(p1, R(w1 * g0)); LDS_ADD(p1 + 1, R(w1 * g1));
            }
        }
        return;
    }
    if (K == KIND_GENERIC) {
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                               // k = (z bit, y bit)
                const int yb = k & 1, zb = k >> 1;
                const float wyz_y = yb ? fy : 1.0f - fy, wyz_z = zb ? fz : 1.0f - fz;
#pragma unroll
                for (int xb = 0; xb < 2; ++xb) {
                    uint32_t loc;
                    const uint32_t h = level_index(P.dense, P.mode, P.size, P.res, cx + xb, cy + yb, cz + zb);
                    if (slice_of(P.map, h, loc) == sl) {
                        const float w = ((1.0f * (xb ? fx : 1.0f - fx)) * wyz_y) * wyz_z;
                        double* p = slice + 2 * loc;
                        LDS_ADD(p, R(w * g0));
                        LDS_ADD(p + 1, R(w * g1));
                    }
                }
            }
        }
        return;
    }
    // KIND_MERGE / KIND_MERGE0: consecutive hits are consecutive samples of a ray; on a coarse level they sit in the same cell for many steps.
    // Sum each equal-cell run (in f32, fixed lane order) with a segmented scan inside 16-lane rows (DPP row shifts: one VALU
    // instruction per value and step); only a run's last lane touches the LDS.  A run that crosses a row boundary simply
    // becomes two adds.
    if (!act) { cx = 0xffffffffu; g0 = 0.f; g1 = 0.f; }
    float v0[8], v1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w = ((1.0f * ((c & 1) ? fx : 1.0f - fx)) * (((c >> 1) & 1) ? fy : 1.0f - fy)) * ((c >> 2) ? fz : 1.0f - fz);
        v0[c] = R(w * g0); v1[c] = R(w * g1);
    }
    const uint32_t pcx = (uint32_t)row_shr_i<1>((int)cx, -1), pcy = (uint32_t)row_shr_i<1>((int)cy, -1), pcz = (uint32_t)row_shr_i<1>((int)cz, -1);
    const bool head = ((lane & 15) == 0) || !act || cx != pcx || cy != pcy || cz != pcz;
    const int nhead = __builtin_amdgcn_update_dpp(1, (int)head, 0x101, 0xf, 0xf, false);          // row_shl:1, row end -> 1
    const bool tail = act && (nhead != 0);
    int hf = head;
    seg_step<1>(v0, v1, hf); seg_step<2>(v0, v1, hf); seg_step<4>(v0, v1, hf); seg_step<8>(v0, v1, hf);
    if (tail) {
        const uint32_t res2 = P.res * P.res, base = cx + cy * P.res + cz * res2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t loc;
            const uint32_t h = K == KIND_MERGE0 ? dense0_index(base, P.res, res2, P.size, c)
                                                   : level_index(P.dense, P.mode, P.size, P.res, cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2));
            if (slice_of(P.map, h, loc) == sl) {
                double* p = slice + 2 * loc;
                LDS_ADD(p, v0[c]);
                LDS_ADD(p + 1, v1[c]);
            }
        }
    }
}

// A pair of batches (lane holds hits i0 and i1): the four gathers of a lane are issued together.
struct Batch {
    Hit h0, h1;
    bool v0, v1;
};

__device__ __forceinline__ Batch load_batch(const int level, const int i0, const bool v0, const int i1, const bool v1,
                                            const float* __restrict__ xyzc, const float* __restrict__ dout, const size_t plane,
                                            const int enc_pairs, const int nl, int32_t* __restrict__ found_inf, const uint32_t diag,
                                            const bool half) {
    Batch b;
    b.v0 = v0; b.v1 = v1;
    b.h0 = load_hit(level, i0, v0, xyzc, dout, plane, enc_pairs, nl, found_inf, diag, half);
    b.h1 = load_hit(level, i1, v1, xyzc, dout, plane, enc_pairs, nl, found_inf, diag, half);
    return b;
}

template <int KIND>
__device__ __forceinline__ void accumulate_batch(const LevelParams P, const uint32_t sl, const bool single, const Batch& b,
                                                 double* __restrict__ slice, int32_t* __restrict__ found_inf) {
    // GradScaler's inf/nan check, where the data passes -- here, not at the load: testing a value the moment it is requested
    // would make the wave wait for the gather it has just issued
    if (found_inf && !(isfinite(b.h0.g0) && isfinite(b.h0.g1) && isfinite(b.h1.g0) && isfinite(b.h1.g1))) *found_inf = 1;
#ifdef NGP_BWD_DIAG
    if (P.diag & 4u) { asm volatile("" :: "v"(b.h0.x), "v"(b.h0.g0), "v"(b.h1.x), "v"(b.h1.g0)); return; }
#endif
    accumulate<KIND>(P, sl, single, b.h0, b.v0, slice);
    accumulate<KIND>(P, sl, single, b.h1, b.v1, slice);
}

// One task: software-pipelined.  The wave's share of the hit bitmap is fetched 64 words (4096 samples) per vector load, one
// super-chunk ahead; the gathers of batch b+1 are issued before batch b is accumulated.
template <int KIND>
__device__ __forceinline__ void bwd_task(const LevelParams P, const int level, const uint32_t sl, const bool single, const int n,
                                         const int rep, const int nrep, const float* __restrict__ xyzc,
                                         const unsigned long long* __restrict__ brow, const float* __restrict__ dout,
                                         const size_t plane, const int enc_pairs, const int nl, double* __restrict__ slice,
                                         uint32_t* __restrict__ q, uint32_t* __restrict__ next_sc, int32_t* __restrict__ found_inf) {
    constexpr bool HALF = (KIND & KIND_HALF) != 0;
    constexpr int K = KIND & 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this replica's sample range in 64-sample words, whole 8-word chunks
    const int n_words = (n + 63) >> 6;
    int chunk = (n_words + nrep - 1) / nrep;
    chunk = (chunk + 7) & ~7;
    const int lo_w = rep * chunk, hi_w = min(n_words, lo_w + chunk);
    Batch pend;
    pend.v0 = pend.v1 = false;
    pend.h0 = pend.h1 = Hit{0.f, 0.f, 0.f, 0.f, 0.f};
    if (single) {
        for (int w0 = lo_w + 2 * wave; w0 < hi_w; w0 += 2 * BW_WAVES) {
            const int i0 = w0 * 64 + lane, i1 = i0 + 64;
            const Batch nxt = load_batch(level, i0, i0 < n, i1, (w0 + 1 < hi_w) && i1 < n, xyzc, dout, plane, enc_pairs, nl, found_inf, P.diag, HALF);
            accumulate_batch<KIND>(P, sl, true, pend, slice, found_inf);
            pend = nxt;
        }
        accumulate_batch<KIND>(P, sl, true, pend, slice, found_inf);
        return;
    }
    int qhead = 0, qlen = 0;
    // super-chunk c = words [lo_w + SCW c, + SCW): lane l < SCW holds word l.  Waves take super-chunks from a shared LDS counter
    // (hit density varies along the sample list; a static deal left the slowest wave 15-30 % behind), one ahead of the one in
    // work.  SCW = 64 words (4096 samples, ~256 hits) where ~6 % of the samples hit (hashed levels); fewer where a larger share
    // hits (dense levels: ~2 of ns slices per sample, replicated over short sample ranges -- a 4096-sample chunk would leave
    // most of the 16 waves idle there, a 256-sample chunk of a 32-slice level would be one exposed load latency per 25 hits).
    int SCW = 64;
    if (K == KIND_MERGE || K == KIND_MERGE0) {
        SCW = 4; while (SCW < 64 && SCW < 2 * (int)P.map.ns) SCW <<= 1;
        // ... and never so wide that the range has fewer pieces than `merge_chunks` (round 6: with 8 sample ranges per slice a 32-slice
        // level's task held 13 pieces of 64 words for its 16 waves)
        if (P.merge_chunks > 0) while (SCW > 4 && (hi_w - lo_w) < P.merge_chunks * SCW) SCW >>= 1;
    }
    const int n_sc = (hi_w - lo_w + SCW - 1) / SCW;
    auto load_words = [&](int c) -> unsigned long long {
        const int w = lo_w + c * SCW + lane;
        return (c < n_sc && lane < SCW && w < hi_w) ? brow[w] : 0ull;
    };
    auto grab = [&]() -> int {
        int c = 0;
        if (lane == 0) c = (int)atomicAdd(next_sc, 1u);
        return __builtin_amdgcn_readfirstlane(c);
    };
    auto drain = [&]() {          // 128 queued hits: issue their gathers, accumulate the batch whose gathers were issued last time
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int i0 = (int)q[(qhead + lane) & (BW_Q - 1)], i1 = (int)q[(qhead + 64 + lane) & (BW_Q - 1)];
        const Batch nxt = load_batch(level, i0, true, i1, true, xyzc, dout, plane, enc_pairs, nl, found_inf, P.diag, HALF);
        accumulate_batch<KIND>(P, sl, false, pend, slice, found_inf);
        pend = nxt;
        __builtin_amdgcn_wave_barrier();
        qhead = (qhead + 128) & (BW_Q - 1); qlen -= 128;
    };
    int sc = grab();
    unsigned long long cur = load_words(sc);
    while (sc < n_sc) {
        const int sc_next = grab();
        const unsigned long long nxtw = load_words(sc_next);
        const int wbase = lo_w + sc * SCW;
        if (K == KIND_MERGE || K == KIND_MERGE0) {
            // word-serial: hits enter the queue in sample order (the run pre-summing needs consecutive samples in consecutive
            // lanes); these levels have few slices, so most bits are set and a word's ~30 instructions buy ~64 hits
            unsigned long long nonzero = __ballot(cur != 0ull);                // which of the 64 words have any hit
            while (nonzero) {
                const int k = __builtin_ctzll(nonzero);
                nonzero &= nonzero - 1;
                const unsigned long long b = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(cur >> 32), k) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)cur, k);
                const bool hit = (b >> lane) & 1ull;
                const int pos = qlen + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
                if (hit) q[(qhead + pos) & (BW_Q - 1)] = (uint32_t)((wbase + k) * 64 + lane);
                qlen += __popcll(b);
                if (qlen >= 128) drain();
            }
        } else {
            // lane-parallel.  The first form gave lane l word l (64 samples); every round each lane with bits left emitted its
            // lowest one.  Rounds = the largest popcount among the words (24 on average: hits come in runs of consecutive
            // samples), ~350 clocks each with four waves per SIMD.  Two cheaper scans were built and measured
            // (profiles/r02_hash_bwd_timeline.txt): unpacking a whole word per lane behind a prefix sum, and TRANSPOSED bitmap
            // words (a run spread over 64 lanes: a third of the rounds, scan -40 us).  Both put the equal-cell runs of the coarser
            // hashed levels into ONE add instruction, where they serialise in the LDS (+30 us) and the gathers lose their spread
            // (+24 us): no net gain, not shipped.  What is shipped:
            // The unit a lane owns is an 8-sample PIECE of a word, and only non-empty pieces get a lane: the ~16 non-empty words
            // of a super-chunk (hits come in runs of ~24 consecutive samples) would otherwise keep 48 lanes idle for
            // max-popcount (~24) rounds of 64-bit bit twiddling.  The non-empty pieces (~50 of 512) are compacted, piece-major
            // (neighbouring lanes = different words = different rays), through the free part of this wave's hit queue; a round
            // then is 32-bit work and there are at most 8 of them.
            const unsigned long long w = cur;
            if (__ballot(w != 0ull) != 0ull) {
                int total = 0, pass = 0;
                do {
                    const int sbase = qhead + 128;                   // ring slots [qhead + 128, + 192): free while qlen < 128
                    int off = 0;
#pragma unroll
                    for (int pc = 0; pc < 8; ++pc) {
                        const uint32_t b8 = (uint32_t)(w >> (8 * pc)) & 0xffu;
                        const unsigned long long nz = __ballot(b8 != 0u);
                        const int dst = off - 64 * pass + __builtin_amdgcn_mbcnt_hi((uint32_t)(nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nz, 0));
                        if (b8 != 0u && (uint32_t)dst < 64u) q[(sbase + dst) & (BW_Q - 1)] = b8 | ((uint32_t)lane << 8) | ((uint32_t)pc << 14);
                        off += __popcll(nz);
                    }
                    total = off;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t d = (lane < total - 64 * pass) ? q[(sbase + lane) & (BW_Q - 1)] : 0u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint32_t bits = d & 0xffu;
                    const uint32_t s0 = (uint32_t)(wbase + (int)((d >> 8) & 63u)) * 64u + 8u * (d >> 14);     // sample of the piece's bit 0
                    unsigned long long live = __ballot(bits != 0u);
                    while (live) {
                        const int pos = qlen + __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0));
                        if (bits != 0u) {
                            q[(qhead + pos) & (BW_Q - 1)] = s0 + (uint32_t)__builtin_ctz(bits);
                            bits &= bits - 1u;
                        }
                        qlen += __popcll(live);
                        if (qlen >= 128) drain();
                        live = __ballot(bits != 0u);
                    }
                    ++pass;
                } while (64 * pass < total);
            }
        }
        if ((K == KIND_MERGE || K == KIND_MERGE0) && P.det && qlen > 0) {
            // deterministic mode: a run-pre-summing batch never spans two super-chunks, so WHICH hits share a 16-lane row (and
            // therefore how their f32 pre-sums round) depends on the chunk's content only, not on which wave took which chunk
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool v0 = lane < qlen, v1 = lane + 64 < qlen;
            const int i0 = v0 ? (int)q[(qhead + lane) & (BW_Q - 1)] : 0, i1 = v1 ? (int)q[(qhead + 64 + lane) & (BW_Q - 1)] : 0;
            const Batch nxt = load_batch(level, i0, v0, i1, v1, xyzc, dout, plane, enc_pairs, nl, found_inf, P.diag, HALF);
            accumulate_batch<KIND>(P, sl, false, pend, slice, found_inf);
            pend = nxt;
            __builtin_amdgcn_wave_barrier();
            qhead = (qhead + 128) & (BW_Q - 1); qlen = 0;
        }
        cur = nxtw; sc = sc_next;
    }
    if (qlen > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool v0 = lane < qlen, v1 = lane + 64 < qlen;
        const int i0 = v0 ? (int)q[(qhead + lane) & (BW_Q - 1)] : 0, i1 = v1 ? (int)q[(qhead + 64 + lane) & (BW_Q - 1)] : 0;
        const Batch nxt = load_batch(level, i0, v0, i1, v1, xyzc, dout, plane, enc_pairs, nl, found_inf, P.diag, HALF);
        accumulate_batch<KIND>(P, sl, false, pend, slice, found_inf);
        pend = nxt;
    }
    accumulate_batch<KIND>(P, sl, false, pend, slice, found_inf);
}

// Queue heads: ctr[x] = tasks taken from the front of XCD x's queue (low 16 bits, by its own workgroups) and from the back (high
// 16 bits, by thieves); one atomic add claims one position, a claim is valid while front + back < length.  Returns the index
// into plan.task or 0xffffffff when every queue is empty.  Called by the whole of WAVE 0 (round 3): lanes 0..7 read the eight
// heads with ONE load and the fullest queue is picked with a wave reduction -- the serial form (one thread, eight dependent
// atomic loads per try) took ~10 us per stolen task, which the per-task timeline never showed because it sat between two tasks.
// `own_empty` (wave-uniform, kept by the caller) skips the doomed attempt on the own queue once it has failed.
__device__ __forceinline__ uint32_t claim_task(const BwdPlan& plan, uint32_t* __restrict__ ctr, uint32_t xcc, bool& own_empty) {
    const int lane = threadIdx.x & 63;
    if (!own_empty) {
        uint32_t old = 0u;
        if (lane == 0) old = atomicAdd(&ctr[xcc], 1u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        const uint32_t f = old & 0xffffu, b = old >> 16;
        if (f + b < plan.xlen[xcc]) return plan.xoff[xcc] + f;
        own_empty = true;
    }
    uint32_t dead = 1u << xcc;
    for (int tries = 0; tries < 7; ++tries) {
        int left = 0;
        if (lane < 8 && !((dead >> lane) & 1u)) {
            const uint32_t c = __atomic_load_n(&ctr[lane], __ATOMIC_RELAXED);
            left = (int)plan.xlen[lane] - (int)((c & 0xffffu) + (c >> 16));
        }
        int key = left > 0 ? (left << 3) | (7 - lane) : 0;                 // fullest queue, lowest index on ties
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) key = max(key, __shfl_xor(key, d, NGP_WAVE));
        key = __builtin_amdgcn_readfirstlane(key);
        if (key <= 0) break;
        const int best = 7 - (key & 7);
        uint32_t old = 0u;
        if (lane == 0) old = atomicAdd(&ctr[best], 0x10000u);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        const uint32_t f = old & 0xffffu, b = old >> 16;
        if (f + b < plan.xlen[best]) return plan.xoff[best] + plan.xlen[best] - 1u - b;
        dead |= 1u << best;
    }
    return 0xffffffffu;
}

struct MlpSlabs {                     // optional extra work of the main launch: sum n_parts weight-gradient slabs into dW (NULL: none)
    const float* parts;
    int n_parts;
    float* dW;
};

// Round 5: the optimizer rides in the flush.  The owner of a NON-replicated slice (nrep == 1: every hashed level of a 2^19-entry
// table) holds the complete, final gradient of its 8192 entries in LDS when its task ends.  Instead of adding it to the gradient
// table -- only for the optimizer launch to read it back, clear it and stream p / m / v (train.py:197-201: 32 B per parameter, a
// 64 us launch at 0.92 of its own roofline) -- the owner applies torch.optim.Adam's update right there: read m, v, p of EVERY entry
// of the slice (the moments keep decaying where g = 0), write p, m, v [+ the 16-bit storage copy].  24 B per parameter instead of 40,
// and the optimizer launch shrinks to the replicated coarse levels + the MLP.  The skip / step decision (GradScaler) therefore has
// to exist BEFORE this launch: the MLP backward flags non-finite d_enc / dW (the scatter-add's own input check is the same
// condition), ngp_train_prologue runs in front of the scatter-add, and the kernel reads SI_SKIP like the optimizer kernels do.
// The arithmetic is adam_table_pass's (ngp_device.h), per entry instead of per float4 group: the group form only differs in which
// all-zero entries it visits, and an all-zero entry (g = m = v = 0) is an exact fixed point -- bit-identical tables, moments and
// copies (tests/test_gpu_flush_adam.py).
struct FlushAdam {
    float* p;                    // fp32 master table [entries * 2]; NULL: no optimizer in the flush
    float* m;
    float* v;
    uint16_t* shadow;            // 16-bit storage copy refreshed with p (ADAM == 2: bf16), or NULL
    const float* sf;             // optimizer state (include/ngp_hip.h): SF_INV_SCALE, SF_LR, SF_BC1, SF_BC2_SQRT as the prologue left them
    const int32_t* si;           // SI_SKIP
    float beta1, beta2, eps;
};

// ... and so does the step's scalar bookkeeping (ngp_train_prologue: skip decision, loss-scale update, learning rate, bias
// corrections).  It has to run between the MLP backward (which raises the inf flag) and the first flush; as a launch of its own
// that is one thread for 6 us plus a 4 us gap in front of a 200 us kernel.  Instead every workgroup evaluates it on a private LDS
// copy of the state when it starts (the same inputs, the same arithmetic: the same result in all 256), the flushes read that copy,
// and the LAST workgroup out -- the one that already resets the queue heads -- publishes it.  No workgroup reads the global state
// after that store: being last out means every other workgroup has finished.
struct StepSchedule {
    float* sf;                   // NULL: the caller ran ngp_train_prologue itself
    int32_t* si;
    float lr0, eta_min;
    int t_max;
    float growth, backoff;
    int growth_interval;
};

template <bool HALF, int ADAM>
__global__ void __launch_bounds__(BW_THREADS) hash_bwd_lds_kernel(const float* __restrict__ xyzc,
                                                                  const unsigned long long* __restrict__ bitmap, size_t wstride,
                                                                  const float* __restrict__ dout, ngp_hash_levels lv, int n,
                                                                  const int32_t* __restrict__ n_dev, int enc_pairs, BwdPlan plan,
                                                                  void* __restrict__ dtable /* f32 pairs; HALF: f16 pairs */,
                                                                  int32_t* __restrict__ found_inf, uint32_t* __restrict__ ctr,
                                                                  unsigned long long* __restrict__ dbg, MlpSlabs mlp, FlushAdam ad_in,
                                                                  StepSchedule sch) {
    __shared__ double slice[2 * BW_SLICE_ENTRIES];
    __shared__ float s_sf[8];
    __shared__ int32_t s_si[8];
    FlushAdam ad = ad_in;
    const bool own_schedule = ADAM != 0 && sch.sf != nullptr;
    if (own_schedule) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s_sf[k] = sch.sf[k]; s_si[k] = sch.si[k]; }
            train_prologue_thread(s_sf, s_si, sch.lr0, sch.eta_min, sch.t_max, ad.beta1, ad.beta2, sch.growth, sch.backoff,
                                  sch.growth_interval);
        }
        ad.sf = s_sf; ad.si = s_si;          // (read by the flushes, all of which sit behind the barrier after the first claim)
    }
    __shared__ uint32_t queues[BW_WAVES * BW_Q];
    __shared__ uint32_t next_sc;
    __shared__ uint32_t s_claim;
    // Epilogue of the PREVIOUS kernel riding at the head of this one (FusedTrainer, one GPU): the MLP backward left its weight
    // gradients as per-block slabs; the first 147 of the persistent workgroups add them up (ngp_device.h: 64 weights each, 16
    // loads in flight per thread, ~2 us) before they take their first task.  As its own launch -- or inside the prologue launch
    // that follows this kernel -- the sum cost 6-8 us of the step; here it is hidden in a 200 us launch whose queues rebalance.
    if (mlp.parts) {
        for (int b = blockIdx.x; b < NGP_MLP_REDUCE_BLOCKS; b += gridDim.x) {
            mlp_dw_reduce_block(mlp.parts, mlp.n_parts, mlp.dW, b);
            __syncthreads();
        }
    }
    const size_t plane = (size_t)n;
    if (n_dev) n = min(n, *n_dev);
    if (n <= 0 && !ADAM) return;          // (ADAM: a step without live samples still decays the moments -- the tasks run on empty ranges)
    if (n < 0) n = 0;
    const int tid = threadIdx.x;
    // PERSISTENT workgroups (one per CU) take tasks from the queue of the XCD they actually run on; a workgroup whose XCD has
    // run dry steals from the BACK of the fullest other queue (the queues are sorted longest task first, so what is stolen
    // are short coarse-level tasks; a stolen task misses the victim XCD's L2 but would otherwise have waited).  A static deal of
    // tasks to blockIdx % 8 left the XCDs 165-193 us busy and the launch 218 us long (profiles/r02_hash_bwd_timeline.txt).
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    // Task loop, two barriers per task: the NEXT task is claimed (global atomic round trips) by wave 0 while the others flush the
    // current slice, the flush leaves every slot it read at zero (no separate 128 KB zero pass per task), and the super-chunk
    // counter is reset before the flush barrier.  (Round 3: -15 us on the launch at ~390 k live samples.)
    double2* s2 = reinterpret_cast<double2*>(slice);
    for (int j = tid; j < BW_SLICE_ENTRIES; j += BW_THREADS) s2[j] = make_double2(0.0, 0.0);
    bool own_empty = false;                                               // (wave 0's: its own XCD queue has run dry)
    if (tid < 64) {
        const uint32_t c0 = claim_task(plan, ctr, xcc, own_empty);
        if (tid == 0) { s_claim = c0; next_sc = 0u; }
    }
    __syncthreads();
    uint32_t claim = s_claim;
  while (claim != 0xffffffffu) {
    const uint32_t task = plan.task[claim];
    unsigned long long t_begin = 0;
    if (dbg) t_begin = wall_clock64();
    const int level = task & 0xf, rep = (task >> 10) & 0x3f, nrep = plan.nrep[level];
    const uint32_t sl = (task >> 4) & 0x3f;
    LevelParams P;
    P.scale = lv.scale[level]; P.res = lv.resolution[level]; P.size = lv.map_size[level]; P.offset = lv.offset[level];
    P.dense = level < lv.begin_fast_hash_level;
    if (P.dense) { const uint64_t r = P.res; P.mode = ((uint64_t)P.size >= r * r * r && r >= 2) ? 0u : 2u; }
    else P.mode = (P.size != 0 && (P.size & (P.size - 1)) == 0) ? 1u : 2u;
    P.map = slice_map(P.size, P.res, P.dense);
    P.diag = plan.diag;
    P.det = plan.det;
    P.merge_chunks = plan.merge_chunks;
    const bool single = P.size <= (uint32_t)BW_SLICE_ENTRIES;           // one slice: every sample is a hit, no bitmap
    const bool merge = (plan.merge_mask >> level) & 1u;
    const bool hashed = !P.dense && P.mode == 1u && P.res < (1u << BW_SLICE_LOG2) && !single;

    unsigned long long t_init = 0;
    if (dbg) t_init = wall_clock64();
    uint32_t* q = queues + (tid >> 6) * BW_Q;
    const unsigned long long* brow = bitmap + ((size_t)level * BW_MAX_SLICES + sl) * wstride;
    if (merge && P.dense && P.mode == 0u) bwd_task<KIND_MERGE0 | (HALF ? KIND_HALF : 0)>(P, level, sl, single, n, rep, nrep, xyzc, brow, dout, plane, enc_pairs, lv.n_levels, slice, q, &next_sc, found_inf);
    else if (merge) bwd_task<KIND_MERGE | (HALF ? KIND_HALF : 0)>(P, level, sl, single, n, rep, nrep, xyzc, brow, dout, plane, enc_pairs, lv.n_levels, slice, q, &next_sc, found_inf);
    else if (hashed) bwd_task<KIND_HASHED | (HALF ? KIND_HALF : 0)>(P, level, sl, single, n, rep, nrep, xyzc, brow, dout, plane, enc_pairs, lv.n_levels, slice, q, &next_sc, found_inf);
    else bwd_task<KIND_GENERIC | (HALF ? KIND_HALF : 0)>(P, level, sl, single, n, rep, nrep, xyzc, brow, dout, plane, enc_pairs, lv.n_levels, slice, q, &next_sc, found_inf);
    unsigned long long t_wave = 0;
    if (dbg) t_wave = wall_clock64();
    __syncthreads();
    unsigned long long t_acc = 0;
    if (dbg) t_acc = wall_clock64();
    if (tid < 64) {                                                       // wave 0 claims the next task, then joins the flush
        const uint32_t c1 = claim_task(plan, ctr, xcc, own_empty);
        if (tid == 0) { s_claim = c1; next_sc = 0u; }                     // (s_claim was read by everybody before the barrier above)
    }
    // flush: the slice owner rounds its f64 image to f32 and adds it into the table gradient -- plain (non-atomic)
    // read-modify-write when it is the only replica, float atomics (coalesced, a few thousand lines) when the level is
    // replicated over sample ranges
    if (HALF) {
        // half2 encoder: the gradient table is fp16 pairs (hash_encoder_half.py:291-306); the f64 sum is rounded to fp16 here, once
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        half2v* dh = reinterpret_cast<half2v*>(dtable) + P.offset;
        for (int j = tid; j < BW_SLICE_ENTRIES; j += BW_THREADS) {
            const double2 a = s2[j];
            if (a.x == 0.0 && a.y == 0.0) continue;
            s2[j] = make_double2(0.0, 0.0);                              // the next task starts from a clean slice
            const uint32_t h = entry_of(P.map, sl, (uint32_t)j);
            half2v val;
            val.x = (_Float16)(float)a.x; val.y = (_Float16)(float)a.y;
            if (nrep == 1) dh[h] = dh[h] + val;
            else if (!(val.x == (_Float16)0 && val.y == (_Float16)0))
                __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2v*)(dh + h), val);
        }
    } else if (ADAM && nrep == 1) {
        // the optimizer in the flush (see FlushAdam): UNR entries per thread and trip with all their loads in flight
        const bool skip = ad.si[SI_SKIP] != 0;
        const float inv_scale = ad.sf[SF_INV_SCALE], step_size = ad.sf[SF_LR] / ad.sf[SF_BC1], bc2_sqrt = ad.sf[SF_BC2_SQRT];
        const float beta1 = ad.beta1, beta2 = ad.beta2, eps = ad.eps;
        float2* p2 = reinterpret_cast<float2*>(ad.p) + P.offset;
        float2* m2 = reinterpret_cast<float2*>(ad.m) + P.offset;
        float2* v2 = reinterpret_cast<float2*>(ad.v) + P.offset;
        uint32_t* sh = reinterpret_cast<uint32_t*>(ad.shadow) + P.offset;
        // ONE trip: a thread owns BW_SLICE_ENTRIES / BW_THREADS = 8 entries and requests the moments and the parameter of all of them
        // before it touches anything (24 eight-byte loads in flight, 48 VGPRs: the task's registers are dead here); the LDS sums are
        // read one entry at a time underneath.  (Two trips of four entries: two exposed round trips per flush, ~2.5 flushes per
        // workgroup and launch.)
        // Measured and NOT kept (scripts/gpu_r05_n.sh): requesting them IN FRONT of the end-of-task barrier, so that the round trip
        // hides in the 5-10 us over which the waves arrive there.  4 of the 8 entries early: scatter-add 218-223 -> 254-260 us; all 8:
        // 297-298 us.  The values stay live across the barrier and the next task's claim, the allocator answers with 26 / 58 spilled
        // registers per lane at the task boundaries, and the early waves' loads compete with the late waves' gathers.
        constexpr int UNR = BW_SLICE_ENTRIES / BW_THREADS;
        static_assert(UNR * BW_THREADS == BW_SLICE_ENTRIES, "one trip covers the slice");
        {
            float2 pi[UNR], mi[UNR], vi[UNR];
            uint32_t h[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int j = tid + k * BW_THREADS;
                h[k] = entry_of(P.map, sl, (uint32_t)j);
                const uint32_t hc = (!skip && h[k] < P.size) ? h[k] : 0u;
                mi[k] = m2[hc]; vi[k] = v2[hc]; pi[k] = p2[hc];
            }
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int j = tid + k * BW_THREADS;
                const double2 a = s2[j];
                if (a.x != 0.0 || a.y != 0.0) s2[j] = make_double2(0.0, 0.0);            // the next task starts from a clean slice
                // what the two-launch path hands the optimizer: 0 + (float)sum (the gradient slot is clear at the start of a step)
                const float gx = 0.0f + (float)a.x, gy = 0.0f + (float)a.y;
                float2 mk = mi[k], vk = vi[k], pk = pi[k];
                const bool in = !skip && h[k] < P.size;
                if (!in || (gx == 0.f && gy == 0.f && mk.x == 0.f && mk.y == 0.f && vk.x == 0.f && vk.y == 0.f)) continue;   // exact fixed point
                // The GradScaler decision of this step was taken on the per-sample d_enc values (MLP backward); a slice SUM that leaves the
                // f32 range anyway must not reach the master table and the moments, where nothing could undo it: the entry is
                // left alone and the overflow is recorded -- the last workgroup raises the inf flag in the state it publishes, so the NEXT
                // step is skipped and the loss scale backs off (the two-launch path would have skipped THIS step: same scale trajectory
                // one step later, no poisoned parameter).
                if (!(isfinite(gx) && isfinite(gy))) { atomicOr(&ctr[9], 1u); continue; }
#define NGP_ADAM1(c, g)                                                       \
                {                                                             \
                    const float gr = (g) * inv_scale;                         \
                    mk.c = mk.c + (gr - mk.c) * (1.0f - beta1);               \
                    vk.c = vk.c * beta2 + gr * gr * (1.0f - beta2);           \
                    const float denom = sqrtf(vk.c) / bc2_sqrt + eps;         \
                    pk.c = pk.c - step_size * (mk.c / denom);                 \
                }
                NGP_ADAM1(x, gx) NGP_ADAM1(y, gy)
#undef NGP_ADAM1
                p2[h[k]] = pk; m2[h[k]] = mk; v2[h[k]] = vk;
                if (ADAM == 2) sh[h[k]] = f32_to_bf16_bits(pk.x) | (f32_to_bf16_bits(pk.y) << 16);
            }
        }
    } else {
    float2* dl = reinterpret_cast<float2*>(reinterpret_cast<float*>(dtable) + 2 * (size_t)P.offset);
    for (int j = tid; j < BW_SLICE_ENTRIES; j += BW_THREADS) {
        const double2 a = s2[j];
        if (a.x == 0.0 && a.y == 0.0) continue;
        s2[j] = make_double2(0.0, 0.0);                                // the next task starts from a clean slice
        const uint32_t h = entry_of(P.map, sl, (uint32_t)j);          // LDS position j -> table entry
        const float vx = (float)a.x, vy = (float)a.y;
        if (nrep == 1) {
            float2 d = dl[h];
            d.x += vx; d.y += vy;
            dl[h] = d;
        } else {
            float* dp = reinterpret_cast<float*>(dl + h);
            if (vx != 0.0f) unsafeAtomicAdd(dp, vx);
            if (vy != 0.0f) unsafeAtomicAdd(dp + 1, vy);
        }
    }
    }
    if (dbg) {          // diagnostics (ngp_hash_bwd_sliced_debug): 100 MHz wall-clock stamps per block + wave 0's own finish time
        if (tid == 0 && claim < (uint32_t)BW_MAX_TASKS) {
            unsigned long long* o = dbg + 8 * (size_t)claim;
            o[0] = task; o[1] = t_begin; o[2] = t_init; o[3] = t_wave; o[4] = t_acc; o[5] = wall_clock64(); o[6] = xcc & 0xf; o[7] = (unsigned long long)n;
        }
    }
    __syncthreads();                      // flush done (slice clean again), next claim visible
    claim = s_claim;
  }
    // the last workgroup to leave resets the queue heads for the next launch
    // (Round 6: no __threadfence() in front of the arrival count.  Nothing the last workgroup READS here was written by plain stores of the
    // others -- the queue heads and the overflow word are only ever touched by device-scope atomics, each of them complete before its
    // workgroup got here (the claims' return values were consumed; the flush's atomicOr sits behind the end-of-task barrier's s_waitcnt) --
    // and what it WRITES is for the next launch, behind this launch's end-of-kernel release.  The fence was an L2 write-back per workgroup
    // of the parameters / moments the flushes had just stored: launch -3.5 us, alternating builds on one box.)
    if (tid == 0) {
        if (atomicAdd(&ctr[8], 1u) == gridDim.x - 1u) {
            for (int x = 0; x < 8; ++x) ctr[x] = 0u;
            ctr[8] = 0u;
            const uint32_t overflow = ADAM ? __atomic_load_n(&ctr[9], __ATOMIC_RELAXED) : 0u;      // a flush saw a non-finite slice sum
            if (ADAM) ctr[9] = 0u;
            if (own_schedule) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { sch.sf[k] = s_sf[k]; sch.si[k] = s_si[k]; }
                if (overflow) sch.si[SI_FOUND_INF] = 1;
            }
        }
    }
}

// ---- host: the task plan -------------------------------------------------------------------------------------------------
// Returns false when the level table does not fit the formulation (F != 2, or a level of more than 64 slices).
// Plan knobs.  Release builds read the environment ONCE (NGP_EXPERIMENT bwd_rep_target / NGP_EXPERIMENT bwd_merge_res / NGP_EXPERIMENT bwd_dense_min_rep tune the
// replication of the dense levels); NGP_EXPERIMENT bwd_knobs_dynamic=1 makes them re-read on every call so that one process can A/B them on
// the same inputs (profiles/microbench).  The diagnostic knobs that change RESULTS (NGP_EXPERIMENT bwd_levels drops levels, NGP_BWD_DIAG
// switches pieces of the kernel off) exist only in -DNGP_BWD_DIAG builds (they used to be honoured by every build).
struct Knobs {
    int rep_target = 48, merge_res = 128, dense_min_rep = 8;
    uint32_t level_mask = 0xffffffffu, diag = 0u;
    int blocks = 0;                            // -DNGP_BWD_DIAG: fewer persistent workgroups (contention experiment)
    bool deterministic = false;                // NGP_BWD_PLAN_DETERMINISTIC: no sample-range replicas (no float atomics), see include/ngp_hip.h
    // NGP_BWD_PLAN_CONCENTRATED: the scene fills a small part of the box (multi-cascade scenes), so the COARSE HASHED levels
    // behave like dense ones -- few hot cells carry most samples, and wherever their hash indices fall, a handful of slice owners get
    // several times the mean (C3, profiles/r05_scatter_timeline_garden.txt: level 5's owners 374 us on average, the slowest ~1.3 ms; the
    // XCDs busy 55 % of the launch).  Levels up to hashed_rep_res then get hashed_rep sample-range replicas per slice: C3's launch
    // 2.49 -> 1.70-1.76 ms with the march beside it, the XCDs busy 75 % (profiles/r05_scatter_timeline_garden.txt, second half).
    // merge_hashed: also the dense levels' run pre-summing up to merge_res -- measured, three A/B pairs, 0.6 % slower: off.
    int hashed_rep_res = 0, hashed_rep = 1;
    bool merge_hashed = false;
    int merge_chunks = 32;                     // NGP_EXPERIMENT bwd_merge_chunks (0: round 5's rule; 16 / 32 / 64 measured alike: -8 us of 253)
    bool operator==(const Knobs& o) const {
        return merge_chunks == o.merge_chunks && rep_target == o.rep_target && merge_res == o.merge_res && dense_min_rep == o.dense_min_rep && level_mask == o.level_mask &&
               diag == o.diag && blocks == o.blocks && deterministic == o.deterministic && hashed_rep_res == o.hashed_rep_res &&
               hashed_rep == o.hashed_rep && merge_hashed == o.merge_hashed;
    }
};
static Knobs read_knobs() {
    Knobs k;
    if (const char* e = ngp_experiment("bwd_rep_target")) k.rep_target = atoi(e) > 0 ? atoi(e) : k.rep_target;
    if (const char* e = ngp_experiment("bwd_merge_res")) k.merge_res = atoi(e);
    if (const char* e = ngp_experiment("bwd_dense_min_rep")) k.dense_min_rep = atoi(e) > 0 ? atoi(e) : k.dense_min_rep;
    if (const char* e = ngp_experiment("bwd_merge_chunks")) k.merge_chunks = atoi(e) >= 0 ? atoi(e) : k.merge_chunks;
#ifdef NGP_BWD_DIAG
    if (const char* e = ngp_experiment("bwd_levels")) k.level_mask = (uint32_t)strtoul(e, nullptr, 0);
    if (const char* e = ngp_experiment("bwd_diag")) k.diag = (uint32_t)atoi(e);
    if (const char* e = ngp_experiment("bwd_blocks")) k.blocks = atoi(e);
#endif
    return k;
}
// the plan modes travel with the level table (ngp_hash_levels.bwd_plan): no mode state in the library
static void apply_modes(Knobs& k, uint32_t plan_bits) {
    k.deterministic = (plan_bits & NGP_BWD_PLAN_DETERMINISTIC) != 0u;
    if (plan_bits & NGP_BWD_PLAN_CONCENTRATED) {
        // (NGP_EXPERIMENT bwd_hashed_rep_res / NGP_EXPERIMENT bwd_hashed_rep / NGP_EXPERIMENT bwd_merge_hashed: A/B overrides of what the mode switches on)
        static const int res = [] { const char* e = ngp_experiment("bwd_hashed_rep_res"); return e ? atoi(e) : 256; }();
        static const int rep = [] { const char* e = ngp_experiment("bwd_hashed_rep"); return e && atoi(e) > 0 ? atoi(e) : 3; }();
        static const bool mh = [] { const char* e = ngp_experiment("bwd_merge_hashed"); return e ? atoi(e) != 0 : false; }();
        k.hashed_rep_res = res; k.hashed_rep = rep; k.merge_hashed = mh;
    } else { k.hashed_rep_res = 0; k.hashed_rep = 1; k.merge_hashed = false; }
}
static Knobs knobs(uint32_t plan_bits) {
    static const bool dynamic = ngp_experiment("bwd_knobs_dynamic") != nullptr;
    static const Knobs fixed = read_knobs();
    Knobs k;
#ifndef NGP_BWD_DIAG
    if (!dynamic) { k = fixed; apply_modes(k, plan_bits); return k; }
#endif
    k = read_knobs();
    apply_modes(k, plan_bits);
    return k;
}

static bool build_plan_with(const ngp_hash_levels& lv, BwdPlan& plan, uint32_t& single_mask, int dense_min_rep, const Knobs& K) {
    if (lv.n_features != 2 || lv.n_levels < 1 || lv.n_levels > NGP_MAX_LEVELS) return false;
    struct Lvl { int level, n_slices, nrep, tasks; };
    Lvl lvls[NGP_MAX_LEVELS];
    single_mask = 0u;
    plan.merge_mask = 0u;
    const int rep_target = K.rep_target;  // a replicated level gets ~ rep_target tasks (see the replica comment below)
    const int merge_res = K.merge_res;    // pre-sum equal-cell runs on levels up to this resolution
    const uint32_t level_mask = K.level_mask;      // diagnostics (-DNGP_BWD_DIAG): only these levels' tasks
    for (int l = 0; l < lv.n_levels; ++l) {
        const uint32_t size = lv.map_size[l];
        const int ns = (int)((size + BW_SLICE_ENTRIES - 1) / BW_SLICE_ENTRIES);
        if (ns < 1 || ns > BW_MAX_SLICES || size % 2 != 0) return false;
        // a level of few slices sees a large share of the samples in every slice (a sample touches 1-2 z-adjacent slices of a
        // dense level, and the scene concentrates in a part of the z range): replicate it over sample ranges so that one task
        // handles no more hits than a hashed level's slice owner (S/16: 4 of 64 slices per sample)
        int nrep = ns >= BW_MAX_SLICES ? 1 : (rep_target + ns / 2) / ns;
        // a dense level's slice load follows the SCENE (an unbounded scene whose content sits in a fraction of the box puts
        // every sample into one or two slices of a 44-slice level: measured 637 us at C3): never fewer than `dense_min_rep`
        // sample ranges per dense slice, so the worst slice is bounded by S / dense_min_rep hits; the owners of the empty
        // slices cost ~3 us each
        if (l < lv.begin_fast_hash_level && ns > 1 && nrep < dense_min_rep) nrep = dense_min_rep;
        // concentrated scenes: a coarse hashed level's hot cells land in a few slices -- sample-range replicas bound the worst owner
        if (l >= lv.begin_fast_hash_level && (int)lv.resolution[l] <= K.hashed_rep_res && nrep < K.hashed_rep) nrep = K.hashed_rep;
        if (nrep < 1) nrep = 1;
        if (nrep > 63) nrep = 63;
        if (K.deterministic) nrep = 1;        // one owner per slice: its flush is a plain read-modify-write (or the optimizer), no float atomics
        lvls[l] = {l, ns, nrep, ((level_mask >> l) & 1u) ? ns * nrep : 0};
        plan.nrep[l] = (uint8_t)nrep;
        if (ns == 1) single_mask |= 1u << l;
        if ((int)lv.resolution[l] <= merge_res && (l < lv.begin_fast_hash_level || K.merge_hashed)) plan.merge_mask |= 1u << l;    // coarse levels: dense ones, hashed ones of a concentrated scene
    }
    for (int l = lv.n_levels; l < NGP_MAX_LEVELS; ++l) plan.nrep[l] = 1;
    plan.diag = K.diag;                   // timing experiments only (-DNGP_BWD_DIAG builds): wrong results
    plan.det = K.deterministic ? 1u : 0u;
    plan.merge_chunks = K.merge_chunks;
    // XCD-aware order (block b runs on XCD b % 8, one 1024-thread block per CU): the owners of one level read the same
    // position / gradient lines, and they only find them in L2 if they run on the same XCD at about the same time (measured:
    // a hashed level's owners take 52 us when the level has an XCD to itself, 115 us when its 64 owners are spread over all
    // eight).  So tasks are dealt to XCDs in same-level chunks of up to 32 (one round of an XCD's 32 CUs), largest first, each
    // chunk to the least loaded XCD.
    static thread_local uint16_t lists[8][BW_MAX_TASKS];
    int len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float load[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto encode = [](int level, int slice, int rep) { return (uint16_t)(level | (slice << 4) | (rep << 10)); };
    auto least = [&]() { int b = 0; for (int x = 1; x < 8; ++x) if (load[x] < load[b]) b = x; return b; };
    // task duration model fitted to the measured timelines (profiles/r02_hash_bwd_timeline.txt; microseconds at 400 k live samples):
    // 3 + 2.7 per 1000 hits on the run-merging path, 3 + 1.9 per 1000 hits on the hashed path; a sample hits 4 of a hashed
    // level's 64 slices, ~1.5 of a contiguous dense level's slices (z and z + 1 planes), ~2.7 of an interleaved one's
    auto cost = [&](const Lvl& L) {
        const float S = 400.0f;                                 // thousands of samples
        if (L.level >= lv.begin_fast_hash_level) return 3.0f + 1.9f * S * 4.0f / (float)L.n_slices / (float)L.nrep;
        const float frac = L.n_slices == 1 ? 1.0f : (L.n_slices < 8 ? 1.5f : 2.7f) / (float)L.n_slices;
        return 3.0f + 2.7f * S * frac / (float)L.nrep;
    };
    int order[NGP_MAX_LEVELS];
    for (int l = 0; l < lv.n_levels; ++l) order[l] = l;
    for (int a = 0; a < lv.n_levels; ++a)                       // heaviest levels first (fine levels first on ties)
        for (int b = a + 1; b < lv.n_levels; ++b) {
            const float cb = lvls[order[b]].tasks * cost(lvls[order[b]]), ca = lvls[order[a]].tasks * cost(lvls[order[a]]);
            if (cb > ca || (cb == ca && order[b] > order[a])) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
        }
    int total = 0;
    for (int a = 0; a < lv.n_levels; ++a) total += lvls[order[a]].tasks;
    if (total > BW_MAX_TASKS - 64) return false;
    for (int pass = 0; pass < 2; ++pass)                        // full chunks first, then the remainders
        for (int a = 0; a < lv.n_levels; ++a) {
            const Lvl& L = lvls[order[a]];
            if (L.tasks == 0) continue;
            const int chunk = L.level < lv.begin_fast_hash_level ? 16 : 32;
            const int full = (L.tasks / chunk) * chunk;
            int t = 0, x = 0;
            for (int s = 0; s < L.n_slices; ++s)
                for (int r = 0; r < L.nrep; ++r, ++t) {
                    const bool in_full = t < full;
                    if ((pass == 0) != in_full) continue;
                    if (pass == 0 ? (t % chunk == 0) : (t == full)) x = least();
                    lists[x][len[x]++] = encode(L.level, s, r);
                    load[x] += cost(L);
                }
        }
    // inside an XCD: longest tasks first (stable: a level's owners stay together), so that what is left at the end -- and what a
    // thief takes from the back -- are the short coarse-level tasks (20-30 us), not 50 us ones
    float lcost[NGP_MAX_LEVELS];
    for (int l = 0; l < lv.n_levels; ++l) lcost[l] = cost(lvls[l]);
    int nb = 0;
    for (int x = 0; x < 8; ++x) {
        for (int a = 1; a < len[x]; ++a) {                      // insertion sort, descending cost
            const uint16_t t = lists[x][a];
            int b = a - 1;
            while (b >= 0 && lcost[lists[x][b] & 0xf] < lcost[t & 0xf]) { lists[x][b + 1] = lists[x][b]; --b; }
            lists[x][b + 1] = t;
        }
        plan.xoff[x] = (uint16_t)nb; plan.xlen[x] = (uint16_t)len[x];
        for (int p = 0; p < len[x]; ++p) plan.task[nb++] = lists[x][p];
    }
    plan.n_blocks = nb < BW_PERSISTENT_BLOCKS ? nb : BW_PERSISTENT_BLOCKS;
    if (K.blocks > 0 && K.blocks < plan.n_blocks) plan.n_blocks = K.blocks;
    return true;
}

// rep_target 48 / at least 8 sample ranges per dense slice: 252 us against 261 us at 64 / 4 (prepass + main, 390 k live samples; sweep
// in profiles/r02_hash_bwd_timeline.txt) -- 25 us dense tasks pack better behind the 45 us hashed ones than 50 us ones.  A level
// table whose plan would not fit the task array gets fewer ranges rather than the float-atomic fallback.
static bool build_plan_uncached(const ngp_hash_levels& lv, BwdPlan& plan, uint32_t& single_mask, const Knobs& K) {
    for (int dense_min_rep = K.dense_min_rep; dense_min_rep >= 1; dense_min_rep >>= 1)
        if (build_plan_with(lv, plan, single_mask, dense_min_rep, K)) return true;
    return false;
}

// The plan only depends on the level table and the knobs: it is built once per (thread, level table) and reused by the prepass
// and the main launch of every step (it used to be rebuilt -- getenv, cost model, O(tasks) sort -- twice per step).
struct PlanCache {
    bool valid = false, ok = false;
    ngp_hash_levels key;
    Knobs knobs;
    BwdPlan plan;
    uint32_t single_mask = 0u;
};
// level_mask / max_blocks: ngp_hash_bwd_sliced_main_levels -- the scatter-add of a level group in a launch of its own, on fewer
// than 256 workgroups when something else (a collective) has to find free CUs beside it
static const BwdPlan* get_plan(const ngp_hash_levels& lv, uint32_t& single_mask, uint32_t level_mask = 0xffffffffu, int max_blocks = 0) {
    static thread_local PlanCache cache[8];                      // two tables alternate in a process that trains and evaluates,
    static thread_local int victim = 0;                          // each possibly split into a few level groups
    Knobs K = knobs(lv.bwd_plan);
    K.level_mask &= level_mask;
    if (max_blocks > 0 && (K.blocks <= 0 || max_blocks < K.blocks)) K.blocks = max_blocks;
    for (PlanCache& c : cache)
        if (c.valid && memcmp(&c.key, &lv, sizeof(lv)) == 0 && c.knobs == K) { single_mask = c.single_mask; return c.ok ? &c.plan : nullptr; }
    PlanCache& c = cache[victim];
    victim = (victim + 1) % 8;
    c.valid = true; c.key = lv; c.knobs = K;
    c.ok = build_plan_uncached(lv, c.plan, c.single_mask, K);
    single_mask = c.single_mask;
    return c.ok ? &c.plan : nullptr;
}

// ---- workspace: compact positions | one hit bit per (level, slice, sample) | queue heads
struct WsLayout {
    size_t ms, words;                          // capacity in samples (a multiple of 512) and in 64-sample bitmap words
    size_t off_bitmap, off_ctr, total;         // bytes
};
static WsLayout ws_layout(const ngp_hash_levels& lv, int n_max) {
    WsLayout w;
    w.ms = ((size_t)n_max + 511) & ~(size_t)511;
    w.words = w.ms / 64;
    w.off_bitmap = w.ms * 3 * sizeof(float);
    w.off_ctr = w.off_bitmap + (size_t)lv.n_levels * BW_MAX_SLICES * w.words * sizeof(unsigned long long);
    w.total = w.off_ctr + BW_CTR_BYTES;
    return w;
}

}  // namespace ngp

using namespace ngp;

static unsigned long long* g_bwd_debug = nullptr;

extern "C" {

// diagnostics: when set to a device buffer of 8 * BW_MAX_TASKS (= 8 * 1536) u64, every task of the main kernel records its task
// word, 100 MHz wall-clock stamps (begin, after LDS init, wave 0 done, all waves done, after flush), its XCC id and the sample count
int ngp_hash_bwd_sliced_debug(void* device_buffer) { g_bwd_debug = (unsigned long long*)device_buffer; return 0; }

// host-side introspection (no GPU needed; tests/test_sliced_plan.py): the task plan the main launch would use for this level
// table.  tasks[k] = level | slice << 4 | replica << 10 for k < return value; XCD x owns tasks[xoff[x] .. xoff[x] + xlen[x]);
// nrep[l] = sample-range replicas per slice of level l; bit l of *merge_mask = run pre-summing on level l; bit l of
// *single_mask = one-slice level (no bitmap).  Returns the number of tasks, or -2 when the table cannot be expressed.
int ngp_hash_bwd_sliced_plan(const ngp_hash_levels* lv, uint16_t* tasks, int max_tasks, uint16_t* xoff, uint16_t* xlen, uint8_t* nrep,
                             uint32_t* merge_mask, uint32_t* single_mask) {
    if (!lv) return -1;
    uint32_t sm = 0;
    const BwdPlan* plan = get_plan(*lv, sm);
    if (!plan) return -2;
    int total = 0;
    for (int x = 0; x < 8; ++x) { total += plan->xlen[x]; if (xoff) xoff[x] = plan->xoff[x]; if (xlen) xlen[x] = plan->xlen[x]; }
    if (tasks) for (int k = 0; k < total && k < max_tasks; ++k) tasks[k] = plan->task[k];
    if (nrep) for (int l = 0; l < NGP_MAX_LEVELS; ++l) nrep[l] = plan->nrep[l];
    if (merge_mask) *merge_mask = plan->merge_mask;
    if (single_mask) *single_mask = sm;
    return total;
}

// bytes of scratch the sliced scatter-add needs for buffers of n_max samples: compact positions + one hit bit per (level, slice,
// sample) + the queue heads (ws_layout above)
long long ngp_hash_bwd_sliced_workspace(const ngp_hash_levels* lv, int n_max) {
    if (!lv || n_max <= 0) return 0;
    return (long long)ws_layout(*lv, n_max).total;
}

// The two halves of ngp_hash_bwd_f32_sliced as separate entry points: the prepass only needs the positions and the live list,
// so a caller can issue it before the MLP backward that produces `dout` (FusedTrainer does; on a second stream underneath that
// kernel it was measured slower: the two share the VALU and the cross-stream join costs ~20 us).
int ngp_hash_bwd_sliced_prep(const float* xyzs, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, const int32_t* live_idx,
                             int normalize, float lo, float hi, void* workspace, long long workspace_bytes, void* stream) {
    if (n_max <= 0) return 0;
    if (!workspace || workspace_bytes < ngp_hash_bwd_sliced_workspace(lv, n_max)) return -1;
    uint32_t single_mask;
    const BwdPlan* plan = get_plan(*lv, single_mask);
    if (!plan) return -2;
    const WsLayout W = ws_layout(*lv, n_max);
    char* base = reinterpret_cast<char*>(workspace);
    float* xyzc = reinterpret_cast<float*>(base);
    unsigned long long* bitmap = reinterpret_cast<unsigned long long*>(base + W.off_bitmap);
    uint32_t* ctr = reinterpret_cast<uint32_t*>(base + W.off_ctr);
    const XyzNorm nm = {normalize, lo, hi};
    const PrepLevels pl = make_prep_levels(*lv);
    int lds_begin = 0;                                                    // ballot form below, LDS form from here on
    while (lds_begin < lv->n_levels && pl.l[lds_begin].map.ns <= 8u) ++lds_begin;
    for (int l = lds_begin; l < lv->n_levels; ++l)
        if (pl.l[l].map.ns <= 8u) return -2;                              // (sizes grow with the level in every table ngp_hash_levels_init makes)
    static const int batch = [] { const char* e = ngp_experiment("prep_batch"); return e ? atoi(e) : 1; }();
#define NGP_PREP(B) hipLaunchKernelGGL(hash_bwd_prep_kernel<B>, dim3(BW_PREP_BLOCKS), dim3(256), 0, (hipStream_t)stream, xyzs, live_idx, pl, \
                                       lv->n_levels, lv->begin_fast_hash_level, lds_begin, n_max, n_dev, nm, W.words, single_mask, xyzc, bitmap, ctr)
    if (batch == 3) NGP_PREP(3); else if (batch == 6) NGP_PREP(6); else if (batch == 12) NGP_PREP(12); else NGP_PREP(1);
#undef NGP_PREP
    NGP_LAUNCH_CHECK();
    return 0;
}

// First level whose slice owners are not replicated (nrep == 1), provided those levels are exactly [first, n_levels): the levels
// the flush can run the optimizer for.  -1 when there is no such suffix.
static int adam_first_level(const ngp_hash_levels& lv, const BwdPlan& plan) {
    int first = lv.n_levels;
    while (first > 0 && plan.nrep[first - 1] == 1) --first;
    for (int l = 0; l < first; ++l)
        if (plan.nrep[l] == 1) return -1;
    return first < lv.n_levels ? first : -1;
}

static int sliced_main(bool half, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                       void* dtable, int32_t* found_inf, const void* workspace, long long workspace_bytes, void* stream,
                       uint32_t level_mask = 0xffffffffu, int max_blocks = 0, MlpSlabs mlp = MlpSlabs{nullptr, 0, nullptr},
                       FlushAdam ad = FlushAdam{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f},
                       StepSchedule sch = StepSchedule{nullptr, nullptr, 0.f, 0.f, 0, 0.f, 0.f, 0}) {
    if (n_max <= 0) return 0;
    if (enc_pairs && !(lv->n_features == 2 && lv->n_levels == 16)) return -1;
    if (!workspace || workspace_bytes < ngp_hash_bwd_sliced_workspace(lv, n_max)) return -1;
    uint32_t single_mask;
    const BwdPlan* plan = get_plan(*lv, single_mask, level_mask, max_blocks);
    if (!plan) return -2;
    if (ad.p && (half || adam_first_level(*lv, *plan) < 0)) return -2;
    const WsLayout W = ws_layout(*lv, n_max);
    const char* base = reinterpret_cast<const char*>(workspace);
    const float* xyzc = reinterpret_cast<const float*>(base);
    const unsigned long long* bitmap = reinterpret_cast<const unsigned long long*>(base + W.off_bitmap);
    uint32_t* ctr = reinterpret_cast<uint32_t*>(const_cast<char*>(base) + W.off_ctr);
    if (plan->n_blocks <= 0) {
        // nothing to scatter (an empty level mask): the slab sum the caller was promised still has to happen
        return mlp.parts ? ngp_mlp_dw_reduce(mlp.parts, mlp.n_parts, mlp.dW, stream) : 0;
    }
#define NGP_BWD_LAUNCH(H, A)                                                                                                        \
    hipLaunchKernelGGL((hash_bwd_lds_kernel<H, A>), dim3(plan->n_blocks), dim3(BW_THREADS), 0, (hipStream_t)stream, xyzc, bitmap,    \
                       W.words, dout, *lv, n_max, n_dev, enc_pairs, *plan, dtable, found_inf, ctr, g_bwd_debug, mlp, ad, sch)
    if (half) NGP_BWD_LAUNCH(true, 0);
    else if (ad.p && ad.shadow) NGP_BWD_LAUNCH(false, 2);
    else if (ad.p) NGP_BWD_LAUNCH(false, 1);
    else NGP_BWD_LAUNCH(false, 0);
#undef NGP_BWD_LAUNCH
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_hash_bwd_sliced_main(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs, float* dtable,
                             int32_t* found_inf, const void* workspace, long long workspace_bytes, void* stream) {
    return sliced_main(false, dout, lv, n_max, n_dev, enc_pairs, dtable, found_inf, workspace, workspace_bytes, stream);
}

// ngp_hash_bwd_sliced_main (half = 0) / _main_f16 (half = 1) with the sum of the MLP backward's weight-gradient slabs
// (ngp_mlp_bwd_live_parts -> ngp_mlp_dw_reduce) done by the head of the same launch: dW[9408] += sum over n_parts slabs.
int ngp_hash_bwd_sliced_main_slabs(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                   void* dtable, int table_is_f16, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                                   const float* mlp_dw_parts, int n_parts, float* mlp_dw, void* stream) {
    if (!mlp_dw_parts || !mlp_dw || n_parts <= 0) return -1;
    return sliced_main(table_is_f16 != 0, dout, lv, n_max, n_dev, enc_pairs, dtable, found_inf, workspace, workspace_bytes, stream,
                       0xffffffffu, 0, MlpSlabs{mlp_dw_parts, n_parts, mlp_dw});
}

// The scatter-add of the levels in `level_mask` only (same prepass, same workspace; the other levels' part of dtable is not
// touched): FusedTrainer's overlapped gradient exchange issues the fine levels first and sends their gradient while the coarse
// levels are still being accumulated.  max_blocks > 0 caps the persistent workgroups (a collective needs CUs to run on).
int ngp_hash_bwd_sliced_main_levels(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                    float* dtable, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                                    unsigned int level_mask, int max_blocks, void* stream) {
    return sliced_main(false, dout, lv, n_max, n_dev, enc_pairs, dtable, found_inf, workspace, workspace_bytes, stream, level_mask,
                       max_blocks);
}

// the half2 encoder's backward (hash_encoder_half.py:163-213) on the same prepass: fp16 gradient table [entries][2], the
// encoder's fp16 arithmetic per contribution, ONE rounding of the owner's f64 sum instead of one per atomic add
int ngp_hash_bwd_sliced_main_f16(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                 uint16_t* dtable_f16, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                                 void* stream) {
    return sliced_main(true, dout, lv, n_max, n_dev, enc_pairs, dtable_f16, found_inf, workspace, workspace_bytes, stream);
}

// Round 5: scatter-add + optimizer.  The levels whose slices are not replicated over sample ranges (every level of 64 slices: the
// hashed levels of a 2^19-entry table) never write `dtable`: their owners apply Adam (torch.optim.Adam arithmetic, exactly
// ngp_adam_all_ex's) to table / table_m / table_v [+ the bf16 copy] when they flush.  The other levels' gradient goes to dtable as
// before; the caller runs ngp_adam_all_ex over floats [0, ngp_hash_bwd_sliced_adam_prefix(lv)) afterwards.  state_f / state_i must
// already hold THIS step's decision (ngp_train_prologue in front of this call; the inf flag comes from the MLP backward, which
// checks the very d_enc values this kernel reads).  mlp_dw_parts / n_parts / mlp_dw: optional slab sum as in _main_slabs.
// Returns -2 when the level table has no such levels (the caller keeps the two-launch path).
long long ngp_hash_bwd_sliced_adam_prefix(const ngp_hash_levels* lv) {
    if (!lv) return -1;
    uint32_t sm;
    const BwdPlan* plan = get_plan(*lv, sm);
    if (!plan) return -2;
    const int first = adam_first_level(*lv, *plan);
    if (first < 0) return -2;
    return (long long)lv->offset[first] * lv->n_features;
}

int ngp_hash_bwd_sliced_main_adam(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                  float* dtable, const void* workspace, long long workspace_bytes, const float* mlp_dw_parts,
                                  int n_parts, float* mlp_dw, float* table, float* table_m, float* table_v, uint16_t* table_bf16,
                                  const float* state_f, const int32_t* state_i, float beta1, float beta2, float eps, void* stream) {
    if (!table || !table_m || !table_v || !state_f || !state_i) return -1;
    const MlpSlabs mlp = (mlp_dw_parts && mlp_dw && n_parts > 0) ? MlpSlabs{mlp_dw_parts, n_parts, mlp_dw} : MlpSlabs{nullptr, 0, nullptr};
    return sliced_main(false, dout, lv, n_max, n_dev, enc_pairs, dtable, nullptr, workspace, workspace_bytes, stream, 0xffffffffu, 0, mlp,
                       FlushAdam{table, table_m, table_v, table_bf16, state_f, state_i, beta1, beta2, eps});
}

// ... and with the step's scalar bookkeeping (ngp_train_prologue's arguments) evaluated inside the launch: state_f / state_i are
// read as the previous step left them (+ the inf flag of this step's MLP backward) and hold this step's decision when the launch
// has finished -- what the remaining optimizer launch (ngp_adam_all_ex over the replicated levels + the MLP) then reads.
int ngp_hash_bwd_sliced_main_adam_step(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                       float* dtable, const void* workspace, long long workspace_bytes, const float* mlp_dw_parts,
                                       int n_parts, float* mlp_dw, float* table, float* table_m, float* table_v, uint16_t* table_bf16,
                                       float* state_f, int32_t* state_i, float lr0, float eta_min, int t_max, float beta1, float beta2,
                                       float eps, float growth, float backoff, int growth_interval, void* stream) {
    if (!table || !table_m || !table_v || !state_f || !state_i || t_max <= 0 || n_max <= 0) return -1;
    const MlpSlabs mlp = (mlp_dw_parts && mlp_dw && n_parts > 0) ? MlpSlabs{mlp_dw_parts, n_parts, mlp_dw} : MlpSlabs{nullptr, 0, nullptr};
    return sliced_main(false, dout, lv, n_max, n_dev, enc_pairs, dtable, nullptr, workspace, workspace_bytes, stream, 0xffffffffu, 0, mlp,
                       FlushAdam{table, table_m, table_v, table_bf16, state_f, state_i, beta1, beta2, eps},
                       StepSchedule{state_f, state_i, lr0, eta_min, t_max, growth, backoff, growth_interval});
}

int ngp_hash_bwd_f32_sliced(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                            const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, float* dtable,
                            int32_t* found_inf, void* workspace, long long workspace_bytes, void* stream) {
    if (n_max <= 0) return 0;
    if (enc_pairs && !(lv->n_features == 2 && lv->n_levels == 16)) return -1;
    const int rc = ngp_hash_bwd_sliced_prep(xyzs, lv, n_max, n_dev, live_idx, normalize, lo, hi, workspace, workspace_bytes, stream);
    if (rc != 0) return rc;                                           // -2: not expressible, the caller falls back to the atomic kernel
    return ngp_hash_bwd_sliced_main(dout, lv, n_max, n_dev, enc_pairs, dtable, found_inf, workspace, workspace_bytes, stream);
}

}  // extern "C"
