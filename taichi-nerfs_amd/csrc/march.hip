// march.hip -- ray-AABB intersection and occupancy-grid ray marching for gfx950.
//
// Replaces modules/intersection.py:8-37 and modules/ray_march.py:8-123,197-268 of the reference.
//
// Design (not a translation of the Taichi launch shape):
//   The reference marches every ray twice in one kernel and packs samples with two global atomics per ray.
//   Here the orbit t_{k+1} = t_k + calc_dt(t_k) is recognised as independent of occupancy (occupied cells and
//   the skip loop both advance by calc_dt), so a batch of consecutive orbit points is probed speculatively --
//   32 lanes per ray, one orbit point each, all bitfield loads independent -- and the reference's "examined /
//   skipped / emitted" logic is then resolved over the batch in parallel (march_batches: prefix popcounts, a rank
//   search per skip, pointer doubling over the chain of examined points).  Emitted (t, dt) pairs go to a per-ray
//   staging row.  Two packings of the result:
//     * march_fused_kernel (what the trainer and the fused render launch): ONE launch; a 16-wave block takes its output
//       range with one atomic add and expands its own rays -- the rays follow each other in block-completion order,
//       as in the reference, whose order is that of its atomic adds (ray_march.py:76-80);
//     * march_count / march_scan / march_write (the operator chain): a prefix sum over the per-ray counts, rays_a in
//       ray order -- what a serial execution of the reference produces.
//   Per ray the samples are bit-identical to the reference's serial march either way (checked against the oracle).
#include "ngp_device.h"
#include <stdlib.h>

namespace ngp {

struct MarchParams {
    int cascades, grid_size;
    uint32_t grid_size3;
    float grid_size_f, grid_size_inv, grid_max;   // G, 1/G, G-1
    float scale, esf, dt_min, dt_max;
    float scale_inv, mb0, mb0_inv;                // 1/scale; mip_bound of cascade 0 = min(2^-1, scale) and its reciprocal
    unsigned long long rng_seed;                  // rng != 0: the jitter of ray r is rng_uniform(rng_seed, r) instead of noise[r]
    int rng;
    long long capacity;                           // one-launch march: rows of the output arrays (samples at or beyond it are dropped,
};                                                // `total` still counts them); 0 = the caller guarantees n_rays * max_samples rows

__host__ inline MarchParams make_march_params(int cascades, int grid_size, float scale, float esf) {
    MarchParams p;
    p.cascades = cascades;
    p.grid_size = grid_size;
    p.grid_size3 = (uint32_t)grid_size * (uint32_t)grid_size * (uint32_t)grid_size;
    p.grid_size_f = (float)grid_size;
    p.grid_size_inv = 1.0f / (float)grid_size;
    p.grid_max = (float)grid_size - 1.0f;
    p.rng_seed = 0ull; p.rng = 0; p.capacity = 0;
    p.scale = scale;
    p.esf = esf;
    p.dt_min = (float)(1.7320508075688772 / 1024);                       // utils.py:15
    p.dt_max = (float)(1.7320508075688772 * 2) * scale / (float)grid_size;  // utils.py:16,56-57
    p.scale_inv = 1.0f / scale;
    p.mb0 = 0.5f < scale ? 0.5f : scale;
    p.mb0_inv = 1.0f / p.mb0;
    return p;
}

struct CellProbe {
    float xyz[3];
    float nxyz[3];
    float mip_bound;
    uint32_t idx;
};

// ray_march.py:46-60 for one orbit point.  CASC1 (one cascade, e.g. Synthetic-NeRF): mip == 0 for every point, so
// mip_bound = min(0.5, scale) and its reciprocal are wave-uniform constants.  Otherwise 1/mip_bound is 2^(1-mip) (exact,
// what the IEEE division of 1 by a power of two returns) or the precomputed 1/scale -- bit-identical, no per-point divide.
template <bool CASC1>
__device__ __forceinline__ void probe_cell(const MarchParams& p, const float o[3], const float d[3], float t, float dt,
                                           CellProbe& c) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c.xyz[k] = o[k] + t * d[k];
    float mip_bound_inv;
    int mip = 0;
    if (CASC1) {
        c.mip_bound = p.mb0;
        mip_bound_inv = p.mb0_inv;
    } else {
        float mx = fmaxf(fmaxf(fabsf(c.xyz[0]), fabsf(c.xyz[1])), fabsf(c.xyz[2]));
        int mip_pos = min(p.cascades - 1, max(0, frexp_bit(mx) + 1));                  // utils.py:78-84
        int mip_dt = min(p.cascades - 1, max(0, frexp_bit(dt * p.grid_size_f)));       // utils.py:87-92
        mip = max(mip_pos, mip_dt);
        const float pw = ldexpf(1.0f, mip - 1);
        const bool use_pw = pw <= p.scale;                                             // min(2^(mip-1), scale)
        c.mip_bound = use_pw ? pw : p.scale;
        mip_bound_inv = use_pw ? ldexpf(1.0f, 1 - mip) : p.scale_inv;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = 0.5f * (c.xyz[k] * mip_bound_inv + 1.0f) * p.grid_size_f;
        c.nxyz[k] = fminf(p.grid_max, fmaxf(0.0f, v));
    }
    // nxyz is clamped into [0, G-1] and never NaN (fminf/fmaxf drop NaNs): the hardware cvt is the truncating cast
    c.idx = (uint32_t)mip * p.grid_size3 + morton3d((uint32_t)c.nxyz[0], (uint32_t)c.nxyz[1], (uint32_t)c.nxyz[2]);
}

// ray_march.py:68-71: t_target of the skip taken from an empty cell
__device__ __forceinline__ float skip_target(const MarchParams& p, const float d[3], const float d_inv[3], float t,
                                             const CellProbe& c) {
    float tmin = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = (((c.nxyz[k] + 0.5f + 0.5f * fsign(d[k])) * p.grid_size_inv * 2.0f - 1.0f) * c.mip_bound - c.xyz[k]) * d_inv[k];
        tmin = k ? fminf(tmin, v) : v;
    }
    return t + fmaxf(0.0f, tmin);
}

// ------------------------------------------------------------------------------------------------------
// a-1  ray-AABB slab test, one lane per ray (intersection.py:22-37)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ray_aabb_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        float scale, int n, float2* __restrict__ hits_t) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float half_size = (scale - (-scale)) / 2.0f;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float o = rays_o[3 * r + k], d = rays_d[3 * r + k];
        float inv_d = 1.0f / d;
        float t_min = (0.0f - half_size - o) * inv_d;
        float t_max = (0.0f + half_size - o) * inv_d;
        float a = fminf(t_min, t_max), b = fmaxf(t_min, t_max);
        t1 = k ? fmaxf(t1, a) : a;
        t2 = k ? fminf(t2, b) : b;
    }
    hits_t[r] = (t2 > 0.0f) ? make_float2(fmaxf(t1, 0.01f), t2) : make_float2(-1.0f, -1.0f);
}

// ------------------------------------------------------------------------------------------------------
// a-2  training march, count + stage.
// MARCH_GROUP lanes cooperate on one ray: the orbit t_{k+1} = t_k + calc_dt(t_k) does not depend on occupancy, so
// the lanes of a group probe G consecutive orbit points at once (cell index, bitfield byte, skip target) and then resolve
// the reference's examined / skipped / emitted logic over the batch IN PARALLEL: the orbit ascends, so every "t_u < x"
// question is a prefix of the batch (a ballot + popcount or a 6-probe binary search over the lanes' t), each point knows
// its successor on the chain of examined points, and log2(G) rounds of pointer doubling mark that chain.  (Rounds 1-2
// replayed the batch serially from LDS broadcasts: ~2000 of the kernel's ~2600 instructions per batch; 109 -> 47 us.)
// 8192 rays give 4096 waves instead of the 128 single-lane-per-ray waves that left 7/8 of the SIMDs idle.
// ------------------------------------------------------------------------------------------------------
constexpr int MARCH_GROUP = 32;
constexpr int MARCH_MAX_COARSE_WORDS = 1024;            // 32 768 coarse blocks: up to 8 cascades of a 128^3 grid
constexpr int ORBIT_BATCH = 8;      // used by the test-time kernel (one lane per ray)

// LDS traffic between the lanes of ONE wave: order it for the compiler and the memory pipeline, no block barrier
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One wave marches 64 / G rays (r = first_ray + lane / G); returns the ray's sample count on all of its lanes and leaves the
// (t, dt) pairs in the ray's staging row.  `chain` = the wave's own 64 / G words of LDS, `coarse_s` the block's copy of the
// coarse occupancy bits; every synchronisation inside is wave-local, so blocks of any number of waves can call it.
template <bool CONST_DT, int G, bool CASC1>
__device__ __forceinline__ int march_rays_of_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                  const float2* __restrict__ hits_t, const uint8_t* __restrict__ bits,
                                                  const float* __restrict__ noise, const MarchParams& p, int max_samples, int n_rays,
                                                  const uint32_t* __restrict__ coarse_s, bool use_coarse, float2* __restrict__ stage,
                                                  unsigned long long* __restrict__ chain, int first_ray, float o[3], float d[3]) {
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G;
    const int r = first_ray + grp;
    const bool has_ray = r < n_rays;
    const int rr = has_ray ? r : 0;
    float d_inv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = rays_o[3 * rr + k]; d[k] = rays_d[3 * rr + k]; d_inv[k] = 1.0f / d[k]; }
    float2 h;
    if (hits_t) h = hits_t[rr];
    else {                                            // fused ray-AABB slab test (intersection.py:22-37), same arithmetic
        const float half_size = (p.scale - (-p.scale)) / 2.0f;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t_min = (0.0f - half_size - o[k]) * d_inv[k], t_max = (0.0f + half_size - o[k]) * d_inv[k];
            const float lo = fminf(t_min, t_max), hi = fmaxf(t_min, t_max);
            a1 = k ? fmaxf(a1, lo) : lo;
            a2 = k ? fminf(a2, hi) : hi;
        }
        h = (a2 > 0.0f) ? make_float2(fmaxf(a1, 0.01f), a2) : make_float2(-1.0f, -1.0f);
    }
    float t1 = h.x;
    const float t2 = h.y;
    const float dt_c = calc_dt(0.0f, p.esf, p.dt_min, p.dt_max);                    // the step when exp_step_factor == 0
    if (t1 >= 0.0f) t1 += calc_dt(t1, p.esf, p.dt_min, p.dt_max) * (p.rng ? rng_uniform(p.rng_seed, (unsigned int)rr) : noise[rr]);      // ray_march.py:39-41
    float t = t1;                                    // group-uniform: first orbit point of the current batch
    int n = 0;                                       // group-uniform: samples emitted so far
    float t_target = -INFINITY;                      // group-uniform: orbit points below this are skipped
    float2* row = stage + (size_t)rr * (size_t)max_samples;
    bool live = has_ray && (0.0f <= t) && (t < t2) && (n < max_samples);             // ray_march.py:46
    while (__any(live)) {
        // every lane walks the whole batch of G orbit steps from the batch base (exact f32 adds, no closed form) and keeps
        // the point it owns (step `sub`), the batch's last point and the next batch's base -- no cross-lane traffic
        float tu = t, t_last = t, tt = t;
        if (CONST_DT) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                tu = (k == sub) ? tt : tu;
                t_last = tt;
                tt += dt_c;
            }
        } else {
            for (int k = 0; k < G; ++k) {
                if (k == sub) tu = tt;
                t_last = tt;
                tt += calc_dt(tt, p.esf, p.dt_min, p.dt_max);
            }
        }
        const float t_next_batch = tt;
        const float dtu = CONST_DT ? dt_c : calc_dt(tu, p.esf, p.dt_min, p.dt_max);
        const float t_after = tu + dtu;
        CellProbe c;
        probe_cell<CASC1>(p, o, d, tu, dtu, c);
        bool occ = live;
        if (use_coarse && occ) { const uint32_t cb = c.idx >> 9; occ = (coarse_s[cb >> 5] >> (cb & 31u)) & 1u; }
        if (occ) occ = (bits[c.idx >> 3] >> (c.idx & 7u)) & 1u;                      // ray_march.py:60-61
        const float targ = skip_target(p, d, d_inv, tu, c);                          // ray_march.py:68-71
        // Fast path: a batch with no occupied point and no skip that reaches past the next orbit point cannot emit.  Every
        // examined point of it leaves a skip target that is already behind the next point, so after the batch the target
        // is either the incoming one (whole batch skipped) or irrelevant (-inf is equivalent).
        const unsigned long long interesting = __ballot(live && (occ || targ > t_after));
        const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << ((grp * G) & 63));
        const bool need = (interesting & gmask) != 0ull;
        if (!need) {
            if (live) {
                if (!(t_last < t2)) live = false;                                    // the orbit left the box inside this batch
                if (!(t_target > t_last)) t_target = -INFINITY;
            }
        }
        if (interesting != 0ull) {
            // Replay of the reference's examined / skipped / emitted logic (ray_march.py:46-74) over the batch, in parallel.
            // The orbit ascends, so "t_u < x" holds on a prefix of the batch and every question below is a population count:
            //   nv  = points still inside the box; e0 = first point not inside the incoming skip;
            //   c_u = first point not inside the skip point u would start  ->  successor of u on the chain of EXAMINED points
            //         j_u = u + 1 (occupied) or max(u + 1, c_u) (empty).
            // The examined set is the chain e0, j(e0), j(j(e0)), ...: log2(G) rounds of pointer doubling mark it in an LDS word.
            const unsigned long long GM = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull);
            const int gbase = grp * G;
            const int gsh = gbase & 63;
            const bool act = live && need;                                            // group-uniform
            const int nv = __popcll((__ballot(tu < t2) >> gsh) & GM);
            const int e0 = __popcll((__ballot(tu < t_target) >> gsh) & GM);
            const unsigned long long occm = (__ballot(occ) >> gsh) & GM;
            int c = 0;
#pragma unroll
            for (int s = G / 2; s >= 1; s >>= 1) {
                const float tv = __shfl(tu, gbase + c + s - 1);
                if (tv < targ) c += s;
            }
            if (__shfl(tu, gbase + G - 1) < targ) c = G;
            int j = occ ? sub + 1 : max(sub + 1, c);
            if (sub == 0) chain[grp] = (e0 < G) ? (1ull << e0) : 0ull;
            bool on_chain = sub == e0;
#pragma unroll
            for (int k = 1; k < G; k <<= 1) {                                         // chain members at distance < k are marked
                wave_sync_lds();
                if (on_chain && j < G) atomicOr(&chain[grp], 1ull << j);
                wave_sync_lds();
                on_chain = (chain[grp] >> sub) & 1ull;
                if (2 * k < G) {
                    const int jj = __shfl(j, gbase + min(j, G - 1));
                    j = (j >= G) ? G : jj;
                }
            }
            const unsigned long long inside = (nv >= 64) ? ~0ull : ((1ull << nv) - 1ull);
            const unsigned long long X = act ? (chain[grp] & inside) : 0ull;          // examined points
            const unsigned long long E = X & occm;                                    // ... that are occupied: emitted, :63-65
            const int cnt = __popcll(E), cap = max_samples - n;                       // cap >= 1 while live
            const int rank = __popcll(E & ((1ull << sub) - 1ull));
            if (((E >> sub) & 1ull) && rank < cap) row[n + rank] = make_float2(tu, dtu);
            const int last = X ? 63 - __clzll(X) : 0;
            const float targ_last = __shfl(targ, gbase + last);
            if (act) {
                if (X) t_target = ((occm >> last) & 1ull) ? -INFINITY : targ_last;    // what the last examined point left behind
                const bool full = cnt >= cap;                                         // n reached max_samples inside the batch
                n = full ? max_samples : n + cnt;
                live = !full && nv == G;                                              // loop head, ray_march.py:46
            }
            wave_sync_lds();
        }
        t = t_next_batch;
    }
    return n;
}


// coarse occupancy: one bit per 8^3 block of cells == per 512 consecutive Morton codes (64 bitfield bytes).
// A clear bit proves the cell empty without touching the bitfield: most batches of a trained scene never issue a
// global load at all, which is what this latency-bound kernel is waiting on.
__device__ __forceinline__ bool load_coarse(const MarchParams& p, const uint32_t* __restrict__ coarse, uint32_t* __restrict__ coarse_s) {
    const int coarse_words = coarse ? (int)((p.grid_size3 >> 9) * (uint32_t)p.cascades + 31u) >> 5 : 0;
    const bool use_coarse = coarse != nullptr && coarse_words <= MARCH_MAX_COARSE_WORDS;
    if (use_coarse) {
        for (int k = threadIdx.x; k < coarse_words; k += blockDim.x) coarse_s[k] = coarse[k];
        __syncthreads();
    }
    return use_coarse;
}

template <bool CONST_DT, int G, bool CASC1>
__global__ void __launch_bounds__(64) march_count_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float2* __restrict__ hits_t,
                                                         const uint8_t* __restrict__ bits, const float* __restrict__ noise,
                                                         MarchParams p, int max_samples, int n_rays,
                                                         const uint32_t* __restrict__ coarse /*nullable*/,
                                                         float2* __restrict__ stage, int32_t* __restrict__ counts) {
    constexpr int GROUPS = 64 / G;
    __shared__ unsigned long long chain[GROUPS];            // bit u: orbit point u of the batch is examined
    __shared__ uint32_t coarse_s[MARCH_MAX_COARSE_WORDS];
    const bool use_coarse = load_coarse(p, coarse, coarse_s);
    float o[3], d[3];
    const int n = march_rays_of_wave<CONST_DT, G, CASC1>(rays_o, rays_d, hits_t, bits, noise, p, max_samples, n_rays, coarse_s, use_coarse,
                                                         stage, chain, blockIdx.x * GROUPS, o, d);
    const int r = blockIdx.x * GROUPS + (int)threadIdx.x / G;
    if (r < n_rays && threadIdx.x % G == 0) counts[r] = n;
}

#ifdef NGP_MARCH_DIAG
// timing builds only (profiles/microbench/march_waves.py): per wave of the one-launch march, 100 MHz stamps (start, march done,
// block done) and the sample count of its first ray
__device__ unsigned long long ngp_march_dbg[4 * 8192];
#endif

// ---- the whole training march in ONE launch: count, allocate, expand -------------------------------------------------------
// A 16-wave block marches 16 * 64 / G rays, takes its output range with one atomic add on a device counter (block-local
// prefix inside it) and expands its rays' staged (t, dt) pairs into xyzs / dirs / deltas / ts itself.  The samples of a ray are
// contiguous and in march order; the RAYS follow each other in the order their blocks finished -- like the reference, whose
// rays take their ranges with atomic adds (ray_march.py:76-80), and unlike the count / scan / write chain above, which packs
// in ray order and costs two more launches (14 + 15-24 us) behind the count.  ctr[0] = allocation counter, ctr[1] = finished
// blocks: the last block publishes total = ctr[0] and clears both, so the pair needs zeroing only once, at allocation.
// Round 5: WAVES per block is a template parameter (16 = rounds 3-4's block).  The per-ray result does not depend on it (every
// synchronisation of march_rays_of_wave is wave-local); what it changes is how the launch shares a CU with the kernels it is put
// underneath: a 4-wave block (8 rays at G = 32) asks for one wave slot per SIMD and 52 VGPRs each, `lds_pad` bytes of dynamic LDS it
// never touches cap how many such blocks a CU takes (ngp_march_train_fused_shaped).
template <bool CONST_DT, int G, bool CASC1, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) march_fused_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           const float2* __restrict__ hits_t, const uint8_t* __restrict__ bits,
                                                           const float* __restrict__ noise, MarchParams p, int max_samples, int n_rays,
                                                           const uint32_t* __restrict__ coarse, float2* __restrict__ stage,
                                                           int32_t* __restrict__ ctr, int32_t* __restrict__ rays_a,
                                                           int32_t* __restrict__ total, float* __restrict__ xyzs,
                                                           float* __restrict__ dirs, float* __restrict__ deltas, float* __restrict__ ts) {
    constexpr int GROUPS = 64 / G, RPB = WAVES * GROUPS;   // rays per block: 32 at G = 32, WAVES = 16
    static_assert(RPB <= 64, "wave 0 scans the block's ray counts with one wave scan");
    __shared__ unsigned long long chain[WAVES][GROUPS];
    __shared__ uint32_t coarse_s[MARCH_MAX_COARSE_WORDS];
    __shared__ int s_off[RPB];
    const bool use_coarse = load_coarse(p, coarse, coarse_s);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane / G, sub = lane % G;
    const int first = blockIdx.x * RPB + wv * GROUPS, r = first + grp;
    float o[3], d[3];
#ifdef NGP_MARCH_DIAG
    const unsigned long long t_w0 = wall_clock64();
#endif
    const int n = march_rays_of_wave<CONST_DT, G, CASC1>(rays_o, rays_d, hits_t, bits, noise, p, max_samples, n_rays, coarse_s, use_coarse,
                                                         stage, chain[wv], first, o, d);
#ifdef NGP_MARCH_DIAG
    if (lane == 0 && blockIdx.x * WAVES + wv < 8192) {
        unsigned long long* q = ngp_march_dbg + 4 * (blockIdx.x * WAVES + wv);
        q[0] = t_w0; q[1] = wall_clock64(); q[3] = (unsigned long long)n;
    }
#endif
    const bool has_ray = r < n_rays;
    if (sub == 0) s_off[wv * GROUPS + grp] = has_ray ? n : 0;
    __threadfence_block();                                  // the staging rows are read back by other lanes below
    __syncthreads();
    if (wv == 0) {
        const int c = lane < RPB ? s_off[lane] : 0;
        const int inc = wave_scan_add_i(c, lane);
        int base = 0;
        if (lane == NGP_WAVE - 1) base = atomicAdd(&ctr[0], inc);
        base = __shfl(base, NGP_WAVE - 1, NGP_WAVE);
        if (lane < RPB) s_off[lane] = base + inc - c;
    }
    __syncthreads();
    if (has_ray) {
        const int start = s_off[wv * GROUPS + grp];
        if (sub == 0) { rays_a[3 * r] = r; rays_a[3 * r + 1] = start; rays_a[3 * r + 2] = n; }
        const float2* row = stage + (size_t)r * (size_t)max_samples;
        for (int k = sub; k < n; k += G) {
            const float2 s = row[k];
            const size_t g = (size_t)start + k;
            if (p.capacity > 0 && (long long)g >= p.capacity) break;                 // (never write past the caller's arrays)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                xyzs[3 * g + a] = o[a] + s.x * d[a];                               // ray_march.py:88
                dirs[3 * g + a] = d[a];
            }
            ts[g] = s.x;
            deltas[g] = s.y;
        }
    }
    // Last block out publishes the total and clears the counters.  No device-scope fence: on gfx950 that is an L2 write-back per
    // block (measured: the kernel 4x slower); nothing but the two counters travels between blocks, both are device-scope atomics
    // on one cache line, and every block's allocation precedes its own "finished" increment in program order.  (the two
    // atomics are issued by different lanes.  The allocation is a RETURNING atomic whose value wave 0 broadcasts before the block
    // barrier above, i.e. it has completed at the L2 before any lane passes that barrier; the "finished" increment is issued after
    // it, and both are executed by the L2 in arrival order.  A release / acquire pair at agent scope would say the same to the
    // memory model at the price of the write-back measured above; tests assert total == sum of the per-ray counts on every case.)
#ifdef NGP_MARCH_DIAG
    if (lane == 0 && blockIdx.x * WAVES + wv < 8192) ngp_march_dbg[4 * (blockIdx.x * WAVES + wv) + 2] = wall_clock64();
#endif
    if (threadIdx.x == 0) {
        if (atomicAdd(&ctr[1], 1) == (int)gridDim.x - 1) {
            total[0] = atomicExch(&ctr[0], 0);
            atomicExch(&ctr[1], 0);
        }
    }
}

// coarse[k] bit = any occupied cell among Morton codes [512 k, 512 k + 512) = bitfield bytes [64 k, 64 k + 64)
__global__ void __launch_bounds__(256) bitfield_coarsen_kernel(const uint4* __restrict__ bits16, int n_coarse,
                                                               uint32_t* __restrict__ coarse) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;           // one lane per coarse block
    bool any = false;
    if (w < n_coarse) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint4 v = bits16[4 * (size_t)w + k]; any |= (v.x | v.y | v.z | v.w) != 0u; }
    }
    const unsigned long long m = __ballot(any);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && w < n_coarse) coarse[w >> 5] = (uint32_t)m;
    if (lane == 32 && w < n_coarse) coarse[w >> 5] = (uint32_t)(m >> 32);
}

// exclusive prefix sum over per-ray counts -> rays_a (ray order) + total.  One 1024-thread block; every thread owns a
// contiguous slice of ceil(n / 1024) rays (one pass to sum it, one block-wide scan of the 1024 sums, one pass to write):
// two barriers in total -- the chunked version's 3 barriers + a dependent load per 1024 rays made this little kernel the
// second-longest link of the side-stream march chain.
__global__ void __launch_bounds__(1024) march_scan_kernel(const int32_t* __restrict__ counts, int n_rays,
                                                          int32_t* __restrict__ rays_a, int32_t* __restrict__ total) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n_rays + 1023) >> 10;
    const int lo = min(tid * per, n_rays), hi = min(lo + per, n_rays);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    const int inc = wave_scan_add_i(sum, lane);
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wv) ? wave_tot[w] : 0;
    int run = woff + inc - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = counts[i];
        rays_a[3 * i] = i; rays_a[3 * i + 1] = run; rays_a[3 * i + 2] = c;
        run += c;
    }
    if (tid == 1023) total[0] = woff + inc;
}

// ---- live-sample list for the backward pass ------------------------------------------------------------------------------
// Only the samples the compositor actually used -- the first vr[r] of ray r, i.e. those in front of the early-termination point
// (volume_train.py:31-47) -- receive a non-zero gradient; everything behind is exact zero.  In a trained scene that is one
// sample in five to ten, so the MLP backward and the hash scatter-add run on a compacted list: live_idx[j] = index of the
// j-th live sample (ray order, sample order), live_total = their number.  Same results; the float atomics only see fewer zeros.
// One launch: every block repeats the (cheap) scan of all per-ray counts -- 8 coalesced loads per thread at 8192 rays -- and
// leaves the offsets in live_off (all blocks write the same values; a block only reads back what it wrote itself), then fills
// the list for its share of the rays, one wave per ray.  (Rounds 1-2: a one-block scan launch + a fill launch, 13.5 + 5.6 us
// on the step's critical path.)
__global__ void __launch_bounds__(1024) live_compact_kernel(const int32_t* __restrict__ rays_a, const int32_t* __restrict__ vr_per_ray,
                                                            int n_rays, int32_t* __restrict__ live_off, int32_t* __restrict__ live_idx,
                                                            int32_t* __restrict__ live_total) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n_rays + 1023) >> 10;
    const int lo = min(tid * per, n_rays), hi = min(lo + per, n_rays);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += vr_per_ray[i];
    const int inc = wave_scan_add_i(sum, lane);
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wv) ? wave_tot[w] : 0;
    int run = woff + inc - sum;
    for (int i = lo; i < hi; ++i) { live_off[i] = run; run += vr_per_ray[i]; }
    if (blockIdx.x == 0 && tid == 1023) live_total[0] = woff + inc;
    __threadfence_block();
    __syncthreads();
    const int share = (n_rays + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n0 = blockIdx.x * share, n1 = min(n0 + share, n_rays);
    for (int n = n0 + wv; n < n1; n += 16) {
        const int ray = rays_a[3 * n], start = rays_a[3 * n + 1];
        const int cnt = vr_per_ray[ray], base = live_off[ray];
        for (int k = lane; k < cnt; k += NGP_WAVE) live_idx[base + k] = start + k;
    }
}

// expansion: one wave per ray, lanes stride over the ray's staged samples (coalesced stores)
__global__ void __launch_bounds__(256) march_write_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const int32_t* __restrict__ rays_a, const float2* __restrict__ stage,
                                                          int max_samples, int n_rays, float* __restrict__ xyzs,
                                                          float* __restrict__ dirs, float* __restrict__ deltas,
                                                          float* __restrict__ ts) {
    int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = lane_id();
    const int start = rays_a[3 * r + 1], cnt = rays_a[3 * r + 2];
    if (cnt == 0) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = rays_o[3 * r + k]; d[k] = rays_d[3 * r + k]; }
    const float2* row = stage + (size_t)r * (size_t)max_samples;
    for (int k = lane; k < cnt; k += NGP_WAVE) {
        float2 s = row[k];
        size_t g = (size_t)start + k;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            xyzs[3 * g + a] = o[a] + s.x * d[a];                                   // ray_march.py:88
            dirs[3 * g + a] = d[a];
        }
        ts[g] = s.x;
        deltas[g] = s.y;
    }
}

// ------------------------------------------------------------------------------------------------------
// a-3  test-time march (ray_march.py:216-268): <= max_samples per alive ray per call, resumes via hits_t.
// Same speculative orbit batching; slot layout n*max_samples+s like the reference.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) march_test_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        float* __restrict__ hits_t, const int64_t* __restrict__ alive,
                                                        const uint8_t* __restrict__ bits, MarchParams p, int max_samples,
                                                        int n_alive, int64_t* __restrict__ ray_indices,
                                                        uint8_t* __restrict__ valid_mask, float* __restrict__ deltas,
                                                        float* __restrict__ ts, int32_t* __restrict__ samples_counter) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int64_t r = alive[n];
    float o[3], d[3], d_inv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = rays_o[3 * r + k]; d[k] = rays_d[3 * r + k]; d_inv[k] = 1.0f / d[k]; }
    float t = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    int s = 0;
    const size_t base = (size_t)n * (size_t)max_samples;
    float t_target = -INFINITY;
    float t_resume = t;
    bool wrote = false;
    bool live = (0.0f < t) && (t < t2) && (s < max_samples);                        // :230 (strict 0 < t)
    while (live) {
        float tb[ORBIT_BATCH], dtb[ORBIT_BATCH];
        uint8_t ob[ORBIT_BATCH];
        uint32_t ib[ORBIT_BATCH];
        float tt = t;
#pragma unroll
        for (int u = 0; u < ORBIT_BATCH; ++u) {
            float dt = calc_dt(tt, p.esf, p.dt_min, p.dt_max);
            CellProbe c;
            probe_cell<false>(p, o, d, tt, dt, c);
            tb[u] = tt; dtb[u] = dt; ib[u] = c.idx;
            ob[u] = bits[c.idx >> 3];
            tt += dt;
        }
#pragma unroll
        for (int u = 0; u < ORBIT_BATCH; ++u) {
            if (!live) break;
            float tu = tb[u];
            if (!(tu < t2)) { live = false; break; }
            if (tu < t_target) continue;
            if (ob[u] & (1u << (ib[u] & 7u))) {                                     // :250-258
                size_t i = base + s;
                ray_indices[i] = r; valid_mask[i] = 1; ts[i] = tu; deltas[i] = dtb[u];
                t_resume = tu + dtb[u]; wrote = true;
                s += 1;
                t_target = -INFINITY;
                if (s >= max_samples) live = false;
            } else {
                CellProbe c;
                probe_cell<false>(p, o, d, tu, dtb[u], c);
                t_target = skip_target(p, d, d_inv, tu, c);
            }
        }
        t = tt;
    }
    if (wrote) hits_t[2 * r] = t_resume;                                            // :257
    samples_counter[n] = s;                                                         // :268
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_ray_aabb(const float* rays_o, const float* rays_d, float scale, int n_rays, float* hits_t, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(ray_aabb_kernel, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, scale,
                       n_rays, (float2*)hits_t);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_bitfield_coarsen(const uint8_t* density_bitfield, int cascades, int grid_size, uint32_t* coarse, void* stream) {
    const long long cells = (long long)cascades * grid_size * grid_size * grid_size;
    if (cells <= 0 || cells % 512 != 0) return -1;
    const int n_coarse = (int)(cells / 512);
    if (n_coarse % 32 != 0) return -1;
    hipLaunchKernelGGL(bitfield_coarsen_kernel, dim3((n_coarse + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)density_bitfield, n_coarse, coarse);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_march_train_count(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                          const float* noise, int cascades, int grid_size, float scale, float exp_step_factor,
                          int max_samples, int n_rays, float* stage, int32_t* counts, void* stream) {
    return ngp_march_train_count_ex(rays_o, rays_d, hits_t, density_bitfield, nullptr, noise, cascades, grid_size, scale,
                                    exp_step_factor, max_samples, n_rays, stage, counts, stream);
}

int ngp_march_train_count_ex(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                             const uint32_t* coarse, const float* noise, int cascades, int grid_size, float scale,
                             float exp_step_factor, int max_samples, int n_rays, float* stage, int32_t* counts, void* stream) {
    if (n_rays <= 0) return 0;
    MarchParams p = make_march_params(cascades, grid_size, scale, exp_step_factor);
    // lanes per ray: more lanes = more resident waves for this latency-bound kernel, at the price of a longer replay when
    // a batch contains occupied cells or real skips.  NGP_EXPERIMENT march_group overrides (16 / 32 / 64) for experiments.
    const char* ge = ngp_experiment("march_group");
    const int group = ge ? atoi(ge) : MARCH_GROUP;
    hipStream_t s = (hipStream_t)stream;
#define NGP_LAUNCH_MARCH(CD, GG, C1)                                                                                           \
    hipLaunchKernelGGL((march_count_kernel<CD, GG, C1>), dim3((n_rays + 64 / GG - 1) / (64 / GG)), dim3(64), 0, s, rays_o,     \
                       rays_d, (const float2*)hits_t, density_bitfield, noise, p, max_samples, n_rays, coarse, (float2*)stage, \
                       counts)
    const bool cd = exp_step_factor == 0.0f, c1 = cascades == 1;
    if (group == 64) {
        if (cd && c1) NGP_LAUNCH_MARCH(true, 64, true); else if (cd) NGP_LAUNCH_MARCH(true, 64, false);
        else if (c1) NGP_LAUNCH_MARCH(false, 64, true); else NGP_LAUNCH_MARCH(false, 64, false);
    } else if (group == 16) {
        if (cd && c1) NGP_LAUNCH_MARCH(true, 16, true); else if (cd) NGP_LAUNCH_MARCH(true, 16, false);
        else if (c1) NGP_LAUNCH_MARCH(false, 16, true); else NGP_LAUNCH_MARCH(false, 16, false);
    } else {
        if (cd && c1) NGP_LAUNCH_MARCH(true, 32, true); else if (cd) NGP_LAUNCH_MARCH(true, 32, false);
        else if (c1) NGP_LAUNCH_MARCH(false, 32, true); else NGP_LAUNCH_MARCH(false, 32, false);
    }
#undef NGP_LAUNCH_MARCH
    NGP_LAUNCH_CHECK();
    return 0;
}

static int march_train_fused(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                             const uint32_t* coarse, const float* noise, bool rng, unsigned long long seed, int cascades, int grid_size,
                             float scale, float exp_step_factor, int max_samples, int n_rays, float* stage, int32_t* ctr, int32_t* rays_a,
                             int32_t* total, float* xyzs, float* dirs, float* deltas, float* ts, void* stream, long long capacity = 0,
                             int block_waves = 16, int lds_pad = 0) {
    if (n_rays <= 0) return 0;
    if (!ctr || !rays_a || !total || (!rng && !noise) || capacity < 0) return -1;
    if ((block_waves != 16 && block_waves != 8 && block_waves != 4) || lds_pad < 0 || lds_pad > 150 * 1024) return -1;
    MarchParams p = make_march_params(cascades, grid_size, scale, exp_step_factor);
    p.rng = rng ? 1 : 0; p.rng_seed = seed; p.capacity = capacity;
    hipStream_t s = (hipStream_t)stream;
    const int rpb = block_waves * (64 / MARCH_GROUP);
#define NGP_LAUNCH_MARCH_W(CD, C1, W)                                                                                               \
    hipLaunchKernelGGL((march_fused_kernel<CD, MARCH_GROUP, C1, W>), dim3((n_rays + rpb - 1) / rpb), dim3(64 * W), (size_t)lds_pad, s,   \
                       rays_o, rays_d, (const float2*)hits_t, density_bitfield, noise, p, max_samples, n_rays, coarse, (float2*)stage,  \
                       ctr, rays_a, total, xyzs, dirs, deltas, ts)
#define NGP_LAUNCH_MARCH(CD, C1)                                                                                                    \
    do { if (block_waves == 16) NGP_LAUNCH_MARCH_W(CD, C1, 16); else if (block_waves == 8) NGP_LAUNCH_MARCH_W(CD, C1, 8);           \
         else NGP_LAUNCH_MARCH_W(CD, C1, 4); } while (0)
    const bool cd = exp_step_factor == 0.0f, c1 = cascades == 1;
    if (cd && c1) NGP_LAUNCH_MARCH(true, true); else if (cd) NGP_LAUNCH_MARCH(true, false);
    else if (c1) NGP_LAUNCH_MARCH(false, true); else NGP_LAUNCH_MARCH(false, false);
#undef NGP_LAUNCH_MARCH
#undef NGP_LAUNCH_MARCH_W
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_march_train_fused(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                          const uint32_t* coarse, const float* noise, int cascades, int grid_size, float scale, float exp_step_factor,
                          int max_samples, int n_rays, float* stage, int32_t* ctr, int32_t* rays_a, int32_t* total, float* xyzs,
                          float* dirs, float* deltas, float* ts, void* stream) {
    return march_train_fused(rays_o, rays_d, hits_t, density_bitfield, coarse, noise, false, 0ull, cascades, grid_size, scale,
                             exp_step_factor, max_samples, n_rays, stage, ctr, rays_a, total, xyzs, dirs, deltas, ts, stream);
}

// ... with output arrays of only `capacity` rows: samples that would land at or beyond it are dropped (never written); total[0]
// still holds the full count, so the caller sees total > capacity
int ngp_march_train_fused_cap(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                              const uint32_t* coarse, const float* noise, int cascades, int grid_size, float scale,
                              float exp_step_factor, int max_samples, int n_rays, long long capacity, float* stage, int32_t* ctr,
                              int32_t* rays_a, int32_t* total, float* xyzs, float* dirs, float* deltas, float* ts, void* stream) {
    return march_train_fused(rays_o, rays_d, hits_t, density_bitfield, coarse, noise, false, 0ull, cascades, grid_size, scale,
                             exp_step_factor, max_samples, n_rays, stage, ctr, rays_a, total, xyzs, dirs, deltas, ts, stream, capacity);
}

// The same march with the per-ray jitter drawn IN the kernel: rng_uniform(seed, ray) (ngp_device.h) -- no noise tensor and no
// generator launch in front of it (FusedTrainer's prefetched march: torch's uniform_ kernel used to sit on the side stream for
// 130-170 us waiting for a free CU under the scatter-add).
int ngp_march_train_fused_rng(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                              const uint32_t* coarse, unsigned long long seed, int cascades, int grid_size, float scale,
                              float exp_step_factor, int max_samples, int n_rays, float* stage, int32_t* ctr, int32_t* rays_a,
                              int32_t* total, float* xyzs, float* dirs, float* deltas, float* ts, void* stream) {
    return march_train_fused(rays_o, rays_d, hits_t, density_bitfield, coarse, nullptr, true, seed, cascades, grid_size, scale,
                             exp_step_factor, max_samples, n_rays, stage, ctr, rays_a, total, xyzs, dirs, deltas, ts, stream);
}

// ... with the launch SHAPE chosen by the caller: block_waves in {4, 8, 16} waves per block (16 = every other entry point), and
// lds_pad bytes of untouched dynamic LDS per block, which cap the blocks a CU takes at once (160 KB / (pad + ~1 KB)).  Per ray the
// result is the other entry points' bit for bit; what changes is the block-completion order the rays' ranges are packed in (rays_a
// says where, as always) and how the launch shares the chip with a kernel running beside it (FusedTrainer's prefetched march).
int ngp_march_train_fused_shaped(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                                 const uint32_t* coarse, const float* noise, unsigned long long seed, int cascades, int grid_size,
                                 float scale, float exp_step_factor, int max_samples, int n_rays, int block_waves, int lds_pad_bytes,
                                 float* stage, int32_t* ctr, int32_t* rays_a, int32_t* total, float* xyzs, float* dirs, float* deltas,
                                 float* ts, void* stream) {
    return march_train_fused(rays_o, rays_d, hits_t, density_bitfield, coarse, noise, noise == nullptr, seed, cascades, grid_size, scale,
                             exp_step_factor, max_samples, n_rays, stage, ctr, rays_a, total, xyzs, dirs, deltas, ts, stream, 0,
                             block_waves, lds_pad_bytes);
}

__global__ void rng_uniform_kernel(unsigned long long seed, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rng_uniform(seed, (unsigned int)i);
}

// out[i] = rng_uniform(seed, i): the jitter ngp_march_train_fused_rng(seed) gives ray i (tests; callers that want it as a tensor)
int ngp_rng_uniform(unsigned long long seed, int n, float* out, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(rng_uniform_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, n, out);
    NGP_LAUNCH_CHECK();
    return 0;
}

#ifdef NGP_MARCH_DIAG
int ngp_march_debug_read(unsigned long long* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ngp_march_dbg), sizeof(unsigned long long) * (size_t)n_words) == hipSuccess ? 0 : -1;
}
#endif

int ngp_march_train_scan(const int32_t* counts, int n_rays, int32_t* rays_a, int32_t* total, void* stream) {
    hipLaunchKernelGGL(march_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n_rays, rays_a, total);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_live_compact(const int32_t* rays_a, const int32_t* vr_per_ray, int n_rays, int32_t* live_off, int32_t* live_idx,
                     int32_t* live_total, void* stream) {
    if (n_rays <= 0) return 0;
    int blocks = (n_rays + 127) / 128;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(live_compact_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, rays_a, vr_per_ray, n_rays, live_off, live_idx,
                       live_total);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_march_train_write(const float* rays_o, const float* rays_d, const int32_t* rays_a, const float* stage,
                          int max_samples, int n_rays, float* xyzs, float* dirs, float* deltas, float* ts, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(march_write_kernel, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, rays_a,
                       (const float2*)stage, max_samples, n_rays, xyzs, dirs, deltas, ts);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_march_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive_indices,
                   const uint8_t* density_bitfield, int cascades, int grid_size, float scale, float exp_step_factor,
                   int max_samples, int n_alive, int64_t* ray_indices, uint8_t* valid_mask, float* deltas, float* ts,
                   int32_t* samples_counter, void* stream) {
    if (n_alive <= 0) return 0;
    MarchParams p = make_march_params(cascades, grid_size, scale, exp_step_factor);
    hipLaunchKernelGGL(march_test_kernel, dim3((n_alive + 63) / 64), dim3(64), 0, (hipStream_t)stream, rays_o, rays_d, hits_t,
                       alive_indices, density_bitfield, p, max_samples, n_alive, ray_indices, valid_mask, deltas, ts,
                       samples_counter);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
