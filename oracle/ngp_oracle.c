/*
 * ngp_oracle.c -- CPU restatement of the reference's Instant-NGP hot-path kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under taichi-nerfs_amd/ may import, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / the timed CPU
 * baseline, never as a product path.
 *
 * Each function restates one Taichi kernel of /root/reference/modules/*.py line by line in strict IEEE
 * binary32 (build with -ffp-contract=off: no FMA contraction), citing the lines it follows.
 * Pinning: the reference has no tests or golden vectors.  The restatement is pinned
 * against the reference's OWN kernel source executed under oracle/ti_shim (a scalar f32 interpreter of the
 * Taichi DSL subset those files use) -- see oracle/gen_golden.py and tests/golden/.
 *
 * Deliberate, documented differences from a live Taichi run:
 *   - sample packing is in RAY ORDER (exclusive prefix sum) instead of atomic arrival order
 *     (ray_march.py:76-81 is nondeterministic);
 *   - per-ray private transmittance in the compositor instead of the shared T[] array whose T[s+1] write
 *     races with the next ray (volume_train.py:34,47); ws of early-terminated samples is defined as 0;
 *   - hash-table gradient is the TRUE gradient (the reference's torch glue returns it doubled);
 *   - level scales come from the host table (ngp_hash_levels) instead of an in-kernel expf.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/ngp_hip.h"

#define ORA_API __attribute__((visibility("default")))

static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* ------------------------------------------------------------------------------------------------
 * binary16 helpers (gcc 11 has no _Float16 on x86): exact conversions, round-to-nearest-even.
 * ---------------------------------------------------------------------------------------------- */
static inline float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(s);
        float v = (float)m * 5.9604644775390625e-08f; /* 2^-24 */
        return (h & 0x8000u) ? -v : v;
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}
static inline uint16_t d2h(double d) { /* round a double once to binary16 (RNE) */
    uint16_t sign = signbit(d) ? 0x8000u : 0;
    double a = fabs(d);
    if (isnan(d)) return sign | 0x7e00u;
    if (a >= 65520.0) return sign | 0x7c00u;            /* rounds to inf */
    if (a < 5.9604644775390625e-08 * 0.5) return sign;   /* < half of min subnormal -> 0 */
    int e; double m = frexp(a, &e);                      /* a = m*2^e, m in [0.5,1) */
    int E = e - 1;                                       /* a = (2m)*2^E */
    if (E < -14) {                                       /* subnormal: units of 2^-24 */
        double q = a * 16777216.0;                       /* a / 2^-24 */
        double r = nearbyint(q);                         /* RNE (default rounding mode) */
        return sign | (uint16_t)r;                       /* may carry into the normal range */
    }
    double q = (2.0 * m - 1.0) * 1024.0;                 /* 10-bit mantissa */
    double r = nearbyint(q);
    uint32_t mant = (uint32_t)r, ex = (uint32_t)(E + 15);
    if (mant == 1024u) { mant = 0; ex += 1; }
    if (ex >= 31) return sign | 0x7c00u;
    return sign | (uint16_t)((ex << 10) | mant);
}
static inline uint16_t f2h(float f) { return d2h((double)f); }

ORA_API void ora_f32_to_f16(const float* in, int n, uint16_t* out) { for (int i = 0; i < n; ++i) out[i] = f2h(in[i]); }
ORA_API void ora_f16_to_f32(const uint16_t* in, int n, float* out) { for (int i = 0; i < n; ++i) out[i] = h2f(in[i]); }

/* ------------------------------------------------------------------------------------------------
 * Level table: modules/hash_encoder.py:183-205 with modules/utils.py:19-42 (host, float64) for sizes;
 * the f32 `scale`/`resolution` follow the in-kernel grid_scale/grid_resolution (hash_encoder.py:73-80).
 * ---------------------------------------------------------------------------------------------- */
ORA_API int ora_hash_levels_init(ngp_hash_levels* lv, double max_params, int levels, double base_res,
                                 double max_res, int features) {
    if (levels < 1 || levels > NGP_MAX_LEVELS) return -1;
    memset(lv, 0, sizeof(*lv));
    double log_b = (levels > 1) ? log(max_res / base_res) / (double)(levels - 1) : 0.0; /* utils.py:31-39 */
    uint64_t offset = 0;
    int bfhl = levels;
    for (int i = 0; i < levels; ++i) {
        double resolution = ceil(base_res * exp((double)i * log_b) - 1.0) + 1.0;   /* utils.py:19-29 */
        double full = resolution * resolution * resolution;                          /* hash_encoder.py:189 */
        double aligned = (double)((int64_t)((full + 8 - 1) / 8) * 8);                /* utils.py:41-42 */
        double size = max_params < aligned ? max_params : aligned;                   /* hash_encoder.py:194 */
        lv->offset[i] = (uint32_t)offset;
        lv->map_size[i] = (uint32_t)size;
        if (full > size && bfhl == levels) bfhl = i;                                 /* hash_encoder.py:201-203 */
        offset += (uint64_t)size;
        /* in-kernel f32: exp_scale = exp(level*log_scale); scale = base*exp_scale - 1 (hash_encoder.py:73-76) */
        float e = expf((float)i * (float)log_b);
        float sc = (float)base_res * e - 1.0f;
        lv->scale[i] = sc;
        lv->resolution[i] = (uint32_t)ceilf(sc) + 1u;                                /* hash_encoder.py:78-80 */
    }
    lv->n_levels = levels;
    lv->n_features = features;
    lv->begin_fast_hash_level = bfhl;
    lv->total_entries = (int32_t)offset;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a-1  ray_aabb_intersect -- modules/intersection.py:8-37
 * ---------------------------------------------------------------------------------------------- */
ORA_API void ora_ray_aabb(const float* rays_o, const float* rays_d, float scale, int n, float* hits_t) {
    const float near_distance = 0.01f;                         /* utils.py:13 */
    const float half_size = (scale - (-scale)) / 2.0f;          /* intersection.py:16-17 */
    const float center = 0.0f;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        float t1 = 0, t2 = 0;
        for (int k = 0; k < 3; ++k) {
            float o = rays_o[3 * r + k], d = rays_d[3 * r + k];
            float inv_d = 1.0f / d;                             /* :24 */
            float t_min = (center - half_size - o) * inv_d;     /* :26 */
            float t_max = (center + half_size - o) * inv_d;     /* :27 */
            float a = fminf(t_min, t_max), b = fmaxf(t_min, t_max); /* :29-30 */
            t1 = k ? fmaxf(t1, a) : a;                          /* :31 */
            t2 = k ? fminf(t2, b) : b;                          /* :32 */
        }
        if (t2 > 0.0f) { hits_t[2 * r] = fmaxf(t1, near_distance); hits_t[2 * r + 1] = t2; } /* :34-35 */
        else { hits_t[2 * r] = -1.0f; hits_t[2 * r + 1] = -1.0f; }                           /* :36-37 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * March primitives -- modules/utils.py:54-107
 * ---------------------------------------------------------------------------------------------- */
#define SQRT3 1.7320508075688772
static inline float calc_dt(float t, float esf, int grid_size, float scale) {
    const float lo = (float)(SQRT3 / 1024);                    /* utils.py:15 */
    const float hi = (float)(SQRT3 * 2) * scale / (float)grid_size; /* utils.py:16,56-57 */
    float x = t * esf;
    return fminf(hi, fmaxf(lo, x));                            /* ti.math.clamp */
}
ORA_API int ora_frexp_bit(float x) {                           /* utils.py:60-75 */
    int exponent = 0;
    if (x != 0.0f) {
        uint32_t bits = f2u(x);
        exponent = (int)((bits & 0x7f800000u) >> 23) - 127;
        bits &= 0x7fffffu;
        bits |= 0x3f800000u;
        float frac = u2f(bits);
        if (frac < 0.5f) exponent -= 1;
        else if (frac > 1.0f) exponent += 1;
    }
    return exponent;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int mip_from_pos(const float* p, int cascades) { /* utils.py:78-84 */
    float mx = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
    int exponent = ora_frexp_bit(mx) + 1;
    return imin(cascades - 1, imax(0, exponent));
}
static inline int mip_from_dt(float dt, int grid_size, int cascades) { /* utils.py:87-92 */
    int exponent = ora_frexp_bit(dt * (float)grid_size);
    return imin(cascades - 1, imax(0, exponent));
}
static inline uint32_t expand_bits(uint32_t v) {               /* utils.py:95-101 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) { /* utils.py:104-107 */
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline int32_t morton3d_invert1(uint32_t x) {           /* utils.py:110-117 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return (int32_t)x;
}
static inline uint32_t f2u_sat(float v) {                      /* ti.cast(f32 -> u32): truncation, saturating */
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}
static inline float fsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

typedef struct { const float *o, *d; float d_inv[3]; int cascades, grid_size; float scale, esf; const uint8_t* bits; } march_ctx;

/* One examined orbit point at `t` (ray_march.py:46-74 / :87-123 / :230-265).
 * Returns 1 if the cell is occupied (caller emits and does t += dt); otherwise advances *t past the cell
 * exactly like the reference's skip loop and returns 0.  xyz/dt are returned for the emit. */
static inline int march_step(const march_ctx* c, float* t, float xyz[3], float* dt_out) {
    const int G = c->grid_size;
    const uint32_t G3 = (uint32_t)G * (uint32_t)G * (uint32_t)G;
    const float grid_size_inv = 1.0f / (float)G;
    float tt = *t;
    for (int k = 0; k < 3; ++k) xyz[k] = c->o[k] + tt * c->d[k];                       /* :46 */
    float dt = calc_dt(tt, c->esf, G, c->scale);                                       /* :47 */
    int mip = imax(mip_from_pos(xyz, c->cascades), mip_from_dt(dt, G, c->cascades));   /* :48-49 */
    float mip_bound = fminf(ldexpf(1.0f, mip - 1), c->scale);                          /* :51 */
    float mip_bound_inv = 1.0f / mip_bound;                                            /* :52 */
    float nxyz[3];
    for (int k = 0; k < 3; ++k) {
        float v = 0.5f * (xyz[k] * mip_bound_inv + 1.0f) * (float)G;                   /* :54-58 */
        nxyz[k] = fminf((float)G - 1.0f, fmaxf(0.0f, v));
    }
    uint32_t idx = (uint32_t)mip * G3 + morton3d(f2u_sat(nxyz[0]), f2u_sat(nxyz[1]), f2u_sat(nxyz[2])); /* :60 */
    int occ = c->bits[idx / 8u] & (1u << (idx % 8u));                                  /* :61 */
    *dt_out = dt;
    if (occ) return 1;
    float tmin = 0;
    for (int k = 0; k < 3; ++k) {                                                      /* :68-69 */
        float v = (((nxyz[k] + 0.5f + 0.5f * fsign(c->d[k])) * grid_size_inv * 2.0f - 1.0f) * mip_bound - xyz[k]) * c->d_inv[k];
        tmin = k ? fminf(tmin, v) : v;
    }
    float t_target = tt + fmaxf(0.0f, tmin);                                           /* :71 */
    tt += calc_dt(tt, c->esf, G, c->scale);                                            /* :72 */
    while (tt < t_target) tt += calc_dt(tt, c->esf, G, c->scale);                      /* :73-74 */
    *t = tt;
    return 0;
}

/* a-2  raymarching_train_kernel -- modules/ray_march.py:8-123; ray-order packing.
 * Pass A counts (for the prefix sum), pass B re-marches and writes, exactly the reference's two passes.
 * Returns the total number of samples; xyzs/dirs/deltas/ts must hold n_rays*max_samples rows worst case
 * OR may be NULL to only count (then rays_a is still filled). */
ORA_API int64_t ora_march_train(const float* rays_o, const float* rays_d, const float* hits_t,
                                const uint8_t* bitfield, const float* noise, int cascades, int grid_size,
                                float scale, float esf, int max_samples, int n_rays,
                                int32_t* rays_a, float* xyzs, float* dirs, float* deltas, float* ts) {
    float* t1s = (float*)malloc(sizeof(float) * (size_t)(n_rays > 0 ? n_rays : 1));
#pragma omp parallel for schedule(dynamic, 64)
    for (int r = 0; r < n_rays; ++r) {
        march_ctx c = { rays_o + 3 * r, rays_d + 3 * r, {0, 0, 0}, cascades, grid_size, scale, esf, bitfield };
        for (int k = 0; k < 3; ++k) c.d_inv[k] = 1.0f / c.d[k];                        /* :32 */
        float t1 = hits_t[2 * r], t2 = hits_t[2 * r + 1];                              /* :34 */
        if (t1 >= 0) { float dt = calc_dt(t1, esf, grid_size, scale); t1 += dt * noise[r]; } /* :39-41 */
        t1s[r] = t1;
        float t = t1; int N = 0;                                                       /* :43-44 */
        while ((0 <= t) & (t < t2) & ((float)N < (float)max_samples)) {                /* :46 */
            float xyz[3], dt;
            if (march_step(&c, &t, xyz, &dt)) { t += dt; N += 1; }                     /* :63-65 */
        }
        rays_a[3 * r] = r; rays_a[3 * r + 2] = N;                                      /* :79-81 */
    }
    int64_t total = 0;                                                                 /* :76 (deterministic) */
    for (int r = 0; r < n_rays; ++r) { rays_a[3 * r + 1] = (int32_t)total; total += rays_a[3 * r + 2]; }
    if (xyzs) {
#pragma omp parallel for schedule(dynamic, 64)
        for (int r = 0; r < n_rays; ++r) {
            march_ctx c = { rays_o + 3 * r, rays_d + 3 * r, {0, 0, 0}, cascades, grid_size, scale, esf, bitfield };
            for (int k = 0; k < 3; ++k) c.d_inv[k] = 1.0f / c.d[k];
            float t2 = hits_t[2 * r + 1];
            float t = t1s[r]; int samples = 0;                                         /* :83-84 */
            const int N = rays_a[3 * r + 2]; const int64_t start = rays_a[3 * r + 1];
            while ((t < t2) & (samples < N)) {                                         /* :86 */
                float xyz[3], dt; float t_here = t;
                if (march_step(&c, &t, xyz, &dt)) {                                    /* :103-114 */
                    int64_t s = start + samples;
                    for (int k = 0; k < 3; ++k) { xyzs[3 * s + k] = xyz[k]; dirs[3 * s + k] = c.d[k]; }
                    ts[s] = t_here; deltas[s] = dt;
                    t += dt; samples += 1;
                }
            }
        }
    }
    free(t1s);
    return total;
}

/* a-3  raymarching_test_kernel -- modules/ray_march.py:197-268 (slot layout n*max_samples+s). */
ORA_API void ora_march_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive,
                            const uint8_t* bitfield, int cascades, int grid_size, float scale, float esf,
                            int max_samples, int n_alive, int64_t* ray_indices, uint8_t* valid_mask,
                            float* deltas, float* ts, int32_t* samples_counter) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int n = 0; n < n_alive; ++n) {
        int64_t r = alive[n];                                                          /* :217 */
        march_ctx c = { rays_o + 3 * r, rays_d + 3 * r, {0, 0, 0}, cascades, grid_size, scale, esf, bitfield };
        for (int k = 0; k < 3; ++k) c.d_inv[k] = 1.0f / c.d[k];
        float t = hits_t[2 * r], t2 = hits_t[2 * r + 1];                               /* :225-226 */
        int s = 0; int64_t base = (int64_t)n * max_samples;                            /* :228-229 */
        while ((0 < t) & (t < t2) & (s < max_samples)) {                               /* :230 */
            float xyz[3], dt; float t_here = t;
            if (march_step(&c, &t, xyz, &dt)) {                                        /* :250-258 */
                int64_t i = base + s;
                ray_indices[i] = r; valid_mask[i] = 1; ts[i] = t_here; deltas[i] = dt;
                t += dt; hits_t[2 * r] = t; s += 1;
            }
        }
        samples_counter[n] = s;                                                        /* :268 */
    }
}

/* ------------------------------------------------------------------------------------------------
 * a-4  hash_encoder_kernel fp32 -- modules/hash_encoder.py:89-143 ; index fns :43-71
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t hash_index(int dense, const uint32_t g[3], uint32_t res, uint32_t map_size) {
    uint32_t r;
    if (dense) {                                   /* under_hash :53-60 */
        uint32_t stride = 1; r = 0;
        for (int i = 0; i < 3; ++i) { r += g[i] * stride; stride *= res; }
    } else {                                       /* fast_hash :43-51 */
        r = (g[0] * 1u) ^ (g[1] * 2654435761u) ^ (g[2] * 805459861u);
    }
    return r % map_size;                           /* :71 */
}
/* corner enumeration shared by fwd/bwd: fills idx[8] (entry index incl. level offset) and w[8]. */
static inline void hash_corners(const float* xyz, const ngp_hash_levels* lv, int level, uint32_t idx[8], float w[8]) {
    float scale = lv->scale[level];
    uint32_t res = lv->resolution[level];
    uint32_t cell[3]; float fr[3];
    for (int k = 0; k < 3; ++k) {
        float pos = xyz[k] * scale + 0.5f;                         /* :108 */
        cell[k] = f2u_sat(floorf(pos));                            /* :109 */
        fr[k] = pos - (float)cell[k];                              /* :110 */
    }
    for (int c = 0; c < 8; ++c) {                                  /* :116-127 */
        float ww = 1.0f; uint32_t g[3];
        for (int d = 0; d < 3; ++d) {
            if ((c & (1 << d)) == 0) { g[d] = cell[d]; ww *= 1.0f - fr[d]; }
            else { g[d] = cell[d] + 1u; ww *= fr[d]; }
        }
        idx[c] = lv->offset[level] + hash_index(level < lv->begin_fast_hash_level, g, res, lv->map_size[level]); /* :129-137 */
        w[c] = ww;
    }
}
ORA_API void ora_hash_fwd_f32(const float* xyzs, const float* table, const ngp_hash_levels* lv, int n, float* out) {
    const int L = lv->n_levels, F = lv->n_features;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int level = 0; level < L; ++level) {
            uint32_t idx[8]; float w[8];
            hash_corners(xyzs + 3 * i, lv, level, idx, w);
            for (int f = 0; f < F; ++f) {
                float acc = 0.0f;
                for (int c = 0; c < 8; ++c) acc += w[c] * table[(size_t)idx[c] * F + f];   /* :139-140 */
                out[(size_t)i * L * F + level * F + f] = acc;                             /* :142-143 */
            }
        }
}
/* corner indices/weights for tests (bit-exact index parity): idx_out [n,L,8] u32, w_out [n,L,8] f32 */
ORA_API void ora_hash_corners(const float* xyzs, const ngp_hash_levels* lv, int n, uint32_t* idx_out, float* w_out) {
    const int L = lv->n_levels;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int level = 0; level < L; ++level)
            hash_corners(xyzs + 3 * i, lv, level, idx_out + ((size_t)i * L + level) * 8, w_out + ((size_t)i * L + level) * 8);
}
/* backward = transpose of the forward gather (what Taichi autodiff of :139-140 produces), true gradient.
 * Accumulates in double per entry, in sample order -> deterministic; rounded to f32 at the end. */
ORA_API void ora_hash_bwd_f32(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n, float* dtable) {
    const int L = lv->n_levels, F = lv->n_features;
#pragma omp parallel for schedule(dynamic, 1)
    for (int level = 0; level < L; ++level) {                      /* levels own disjoint table slices */
        size_t cnt = (size_t)lv->map_size[level] * F;
        double* acc = (double*)calloc(cnt, sizeof(double));
        size_t base = (size_t)lv->offset[level] * F;
        for (int i = 0; i < n; ++i) {
            uint32_t idx[8]; float w[8];
            hash_corners(xyzs + 3 * i, lv, level, idx, w);
            for (int c = 0; c < 8; ++c)
                for (int f = 0; f < F; ++f)
                    acc[(size_t)idx[c] * F + f - base] += (double)(w[c] * dout[(size_t)i * L * F + level * F + f]);
        }
        for (size_t k = 0; k < cnt; ++k) dtable[base + k] += (float)acc[k];
        free(acc);
    }
}

/* Throughput variant used ONLY by bench.py's cpu_baseline leg: parallel over samples with omp atomics
 * (summation order is then nondeterministic, like the reference's atomic_add on the CUDA/CPU backends). */
ORA_API void ora_hash_bwd_f32_atomic(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n, float* dtable) {
    const int L = lv->n_levels, F = lv->n_features;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int level = 0; level < L; ++level) {
            uint32_t idx[8]; float w[8];
            hash_corners(xyzs + 3 * i, lv, level, idx, w);
            for (int c = 0; c < 8; ++c)
                for (int f = 0; f < F; ++f) {
                    float v = w[c] * dout[(size_t)i * L * F + level * F + f];
#pragma omp atomic
                    dtable[(size_t)idx[c] * F + f] += v;
                }
        }
}

/* ------------------------------------------------------------------------------------------------
 * a-5  half encoder -- modules/hash_encoder_half.py:112-161 (fwd), :164-213 (bwd)
 * table/out are (f16,f16) pairs; pos/scale/w are f32; accumulate in f16 (:159).
 * NB :133 subtracts cast(pos_grid, f16) from the f32 pos (data_type is f16 in that file).
 * ---------------------------------------------------------------------------------------------- */
static inline void hash_corners_half(const float* xyz, const ngp_hash_levels* lv, int level, uint32_t idx[8], float w[8]) {
    float scale = lv->scale[level];
    uint32_t res = lv->resolution[level];
    uint32_t cell[3]; float fr[3];
    for (int k = 0; k < 3; ++k) {
        float pos = xyz[k] * scale + 0.5f;                         /* :131 */
        cell[k] = f2u_sat(floorf(pos));                            /* :132 */
        fr[k] = pos - h2f(f2h((float)cell[k]));                    /* :133: u32 -> f16 (inexact above 2048) */
    }
    for (int c = 0; c < 8; ++c) {
        float ww = 1.0f; uint32_t g[3];
        for (int d = 0; d < 3; ++d) {
            if ((c & (1 << d)) == 0) { g[d] = cell[d]; ww *= 1.0f - fr[d]; }
            else { g[d] = cell[d] + 1u; ww *= fr[d]; }
        }
        idx[c] = lv->offset[level] + hash_index(level < lv->begin_fast_hash_level, g, res, lv->map_size[level]);
        w[c] = ww;
    }
}
ORA_API void ora_hash_fwd_f16(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n, uint16_t* out) {
    const int L = lv->n_levels, F = lv->n_features;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int level = 0; level < L; ++level) {
            uint32_t idx[8]; float w[8];
            hash_corners_half(xyzs + 3 * i, lv, level, idx, w);
            for (int f = 0; f < F; ++f) {
                uint16_t acc = 0;
                for (int c = 0; c < 8; ++c) {
                    uint16_t term = f2h(w[c] * h2f(table[(size_t)idx[c] * F + f]));   /* cast(w*table, f16) :159 */
                    acc = d2h((double)h2f(acc) + (double)h2f(term));                  /* f16 add, single rounding */
                }
                out[((size_t)i * L + level) * F + f] = acc;                          /* :161 */
            }
        }
}
/* :200-213: hash_grad[idx] += f16(w*dout) per corner, skipped when dout or the product is all-zero.
 * The reference does this with f16x2 atomics (order-dependent rounding); the oracle accumulates exactly in
 * double and rounds once, i.e. it is the limit the f16 atomics approximate -- compare with tolerance. */
ORA_API void ora_hash_bwd_f16(const float* xyzs, const uint16_t* dout, const ngp_hash_levels* lv, int n, float* dtable_f32) {
    const int L = lv->n_levels, F = lv->n_features;
#pragma omp parallel for schedule(dynamic, 1)
    for (int level = 0; level < L; ++level) {
        size_t cnt = (size_t)lv->map_size[level] * F;
        double* acc = (double*)calloc(cnt, sizeof(double));
        size_t base = (size_t)lv->offset[level] * F;
        for (int i = 0; i < n; ++i) {
            uint32_t idx[8]; float w[8];
            hash_corners_half(xyzs + 3 * i, lv, level, idx, w);
            for (int c = 0; c < 8; ++c)
                for (int f = 0; f < F; ++f) {
                    float g = h2f(dout[((size_t)i * L + level) * F + f]);
                    acc[(size_t)idx[c] * F + f - base] += (double)h2f(f2h(w[c] * g));
                }
        }
        for (size_t k = 0; k < cnt; ++k) dtable_f32[base + k] += (float)acc[k];
        free(acc);
    }
}

/* The same kernel (:200-213) in the ONE order the reference's source defines when its struct-for runs serially (sample-major,
 * then level, then corner -- the order oracle/ti_shim executes it in): every `hash_grad[idx] += w_grad_dy_temp` is an f16 vector
 * add rounded per feature, skipped when the output gradient or the f16 product vector is all-zero (:209-212).  Bit-exact against
 * the reference-executed fixtures (tests/golden/ref_hash_f16*.npz) on EVERY row; `count` (optional) receives the number of
 * contributions per table row, so tests can pick the rows whose value does not depend on the accumulation order. */
ORA_API void ora_hash_bwd_f16_serial(const float* xyzs, const uint16_t* dout, const ngp_hash_levels* lv, int n, uint16_t* dtable_h,
                                     int32_t* count) {
    const int L = lv->n_levels, F = lv->n_features;
    for (int i = 0; i < n; ++i)
        for (int level = 0; level < L; ++level) {
            const uint16_t* g = dout + ((size_t)i * L + level) * F;
            int any_g = 0;
            for (int f = 0; f < F; ++f) any_g |= (h2f(g[f]) != 0.0f);
            if (!any_g) continue;                                                  /* :209 */
            uint32_t idx[8]; float w[8];
            hash_corners_half(xyzs + 3 * i, lv, level, idx, w);
            for (int c = 0; c < 8; ++c) {
                uint16_t prod[16]; int any_p = 0;
                for (int f = 0; f < F; ++f) { const float p32 = w[c] * h2f(g[f]); any_p |= (p32 != 0.0f); prod[f] = f2h(p32); }   /* :210; the test
                of :211 sees the f32 product, the cast to the table's f16 happens inside the atomic add */
                if (!any_p) continue;                                              /* :211 */
                for (int f = 0; f < F; ++f) {
                    uint16_t* d = dtable_h + (size_t)idx[c] * F + f;
                    *d = d2h((double)h2f(*d) + (double)h2f(prod[f]));             /* :212, f16 add, single rounding */
                }
                if (count) count[idx[c]] += 1;
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * a-6  dir_encoder -- modules/spherical_harmonics.py:16-42 (literal forms kept for rounding)
 * ---------------------------------------------------------------------------------------------- */
ORA_API void ora_sh16_fwd(const float* dirs, int n, float* e) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        float* o = e + 16 * (size_t)i;
        o[0] = 0.28209479177387814f;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
}
/* analytic Jacobian^T * dout (what kernel.grad at spherical_harmonics.py:92 computes), in double. */
ORA_API void ora_sh16_bwd(const float* dirs, const float* dout, int n, float* ddirs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        const float* g = dout + 16 * (size_t)i;
        const double c1 = 0.48860251190291987, c2 = 1.0925484305920792, c6 = 0.94617469575755997,
                     c8 = 0.54627421529603959, c9 = 0.59004358992664352, c10 = 2.8906114426405538,
                     c11 = 0.45704579946446572, c12 = 0.3731763325901154, c14 = 1.4453057213202769;
        double dx = 0, dy = 0, dz = 0;
        dy += -c1 * g[1]; dz += c1 * g[2]; dx += -c1 * g[3];
        dx += c2 * y * g[4]; dy += c2 * x * g[4];
        dy += -c2 * z * g[5]; dz += -c2 * y * g[5];
        dz += 2 * c6 * z * g[6];
        dx += -c2 * z * g[7]; dz += -c2 * x * g[7];
        dx += 2 * c8 * x * g[8]; dy += -2 * c8 * y * g[8];
        dx += c9 * y * (-6 * x) * g[9]; dy += c9 * (-3 * x * x + 3 * y * y) * g[9];
        dx += c10 * y * z * g[10]; dy += c10 * x * z * g[10]; dz += c10 * x * y * g[10];
        dy += c11 * (1 - 5 * z * z) * g[11]; dz += c11 * y * (-10 * z) * g[11];
        dz += c12 * (15 * z * z - 3) * g[12];
        dx += c11 * (1 - 5 * z * z) * g[13]; dz += c11 * x * (-10 * z) * g[13];
        dx += c14 * z * 2 * x * g[14]; dy += -c14 * z * 2 * y * g[14]; dz += c14 * (x * x - y * y) * g[14];
        dx += c9 * (-3 * x * x + 3 * y * y) * g[15]; dy += c9 * x * 6 * y * g[15];
        ddirs[3 * i] = (float)dx; ddirs[3 * i + 1] = (float)dy; ddirs[3 * i + 2] = (float)dz;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a-7  volume_rendering_kernel -- modules/volume_train.py:22-48 (per-ray private T)
 * rgbs: f32 [S,3] (pass f16 data pre-converted; the fp16 case only changes input rounding).
 * ---------------------------------------------------------------------------------------------- */
ORA_API void ora_composite_train_fwd(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                     const int32_t* rays_a, float T_threshold, int n_rays,
                                     int32_t* total_samples, float* opacity, float* depth, float* rgb, float* ws) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int n = 0; n < n_rays; ++n) {
        int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];   /* :23-25 */
        float r0 = 0, r1 = 0, r2 = 0, dep = 0, op = 0; int cnt = 0;                       /* :27-32 */
        float T = 1.0f;                                                                   /* :34 */
        for (int j = 0; j < N; ++j) {
            int s = start + j;
            if (T > T_threshold) {                                                        /* :38 */
                float a = 1.0f - expf(-sigmas[s] * deltas[s]);                            /* :39 */
                float w = a * T;                                                          /* :40 */
                r0 += w * rgbs[3 * s]; r1 += w * rgbs[3 * s + 1]; r2 += w * rgbs[3 * s + 2]; /* :41-43 */
                dep += w * ts[s]; op += w; ws[s] = w;                                     /* :44-46 */
                T = T * (1.0f - a);                                                       /* :47 */
                cnt += 1;                                                                 /* :48 */
            } else ws[s] = 0.0f;       /* reference leaves these uninitialised (torch.empty, :91-94) */
        }
        rgb[3 * ray_idx] = r0; rgb[3 * ray_idx + 1] = r1; rgb[3 * ray_idx + 2] = r2;
        depth[ray_idx] = dep; opacity[ray_idx] = op; total_samples[ray_idx] = cnt;
    }
}
/* Closed-form backward of the loop above, evaluated in double on the f32
 * forward quantities.  dL_dws may be NULL. */
ORA_API void ora_composite_train_bwd(const float* dL_dop, const float* dL_ddep, const float* dL_drgb, const float* dL_dws,
                                     const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                     const int32_t* rays_a, float T_threshold, int n_rays,
                                     float* dL_dsigmas, float* dL_drgbs) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int n = 0; n < n_rays; ++n) {
        int ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1], N = rays_a[3 * n + 2];
        double gr[3] = { dL_drgb[3 * ray_idx], dL_drgb[3 * ray_idx + 1], dL_drgb[3 * ray_idx + 2] };
        double gd = dL_ddep[ray_idx], go = dL_dop[ray_idx];
        /* forward totals */
        double R[3] = {0, 0, 0}, D = 0, O = 0, W = 0; float T = 1.0f; int M = 0;
        for (int j = 0; j < N; ++j) {
            int s = start + j;
            if (!(T > T_threshold)) break;
            float a = 1.0f - expf(-sigmas[s] * deltas[s]); float w = a * T;
            for (int k = 0; k < 3; ++k) R[k] += (double)w * rgbs[3 * s + k];
            D += (double)w * ts[s]; O += w; if (dL_dws) W += (double)dL_dws[s] * w;
            T = T * (1.0f - a); M = j + 1;
        }
        double rb[3] = {0, 0, 0}, db = 0, wb = 0; T = 1.0f;
        for (int j = 0; j < N; ++j) {
            int s = start + j;
            if (j >= M) { dL_dsigmas[s] = 0; for (int k = 0; k < 3; ++k) dL_drgbs[3 * s + k] = 0; continue; }
            float a = 1.0f - expf(-sigmas[s] * deltas[s]); float w = a * T; float Tp = T * (1.0f - a);
            for (int k = 0; k < 3; ++k) rb[k] += (double)w * rgbs[3 * s + k];
            db += (double)w * ts[s]; double gws = dL_dws ? dL_dws[s] : 0.0; wb += gws * w;
            double acc = 0;
            for (int k = 0; k < 3; ++k) { acc += gr[k] * ((double)rgbs[3 * s + k] * Tp - (R[k] - rb[k])); dL_drgbs[3 * s + k] = (float)(gr[k] * w); }
            acc += gd * ((double)ts[s] * Tp - (D - db));
            acc += go * (1.0 - O);
            acc += gws * Tp - (W - wb);
            dL_dsigmas[s] = (float)((double)deltas[s] * acc);
            T = Tp;
        }
    }
}

/* a-8  composite_test -- modules/volume_render_test.py:18-54 */
ORA_API void ora_composite_test(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                                const int64_t* pack_info, int64_t* alive, float T_threshold, int n_alive,
                                float* opacity, float* depth, float* rgb) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int n = 0; n < n_alive; ++n) {
        int64_t start = pack_info[2 * n], steps = pack_info[2 * n + 1], r = alive[n];   /* :19-21 */
        if (steps == 0) { alive[n] = -1; continue; }                                   /* :22-23 */
        float T = 1.0f - opacity[r];                                                   /* :25 */
        float c0 = 0, c1 = 0, c2 = 0, dep = 0, op = 0;
        for (int64_t j = 0; j < steps; ++j) {
            int64_t s = start + j;
            float delta = deltas[s];
            float a = 1.0f - expf(-sigmas[s] * delta);                                 /* :33-34 */
            float w = a * T;                                                           /* :36 */
            c0 += w * rgbs[3 * s]; c1 += w * rgbs[3 * s + 1]; c2 += w * rgbs[3 * s + 2]; /* :41 */
            dep += w * ts[s]; op += w;                                                 /* :42-43 */
            T *= 1.0f - a;                                                             /* :44 */
            if (T <= T_threshold) { alive[n] = -1; break; }                            /* :46-48 */
        }
        rgb[3 * r] += c0; rgb[3 * r + 1] += c1; rgb[3 * r + 2] += c2;                  /* :50-52 */
        depth[r] += dep; opacity[r] += op;                                             /* :53-54 */
    }
}

/* f-4 get_rays -- datasets/ray_utils.py:51-80: rays_d = directions @ c2w[:, :3].T (per-ray or single pose), rays_o = c2w[..., 3]
 * expanded.  The reference's matmul leaves the summation order to the BLAS; restated left to right, no contraction. */
static void ray_from_pose(const float* c2w, const float* dir, float* o, float* d) {
    for (int i = 0; i < 3; ++i) {
        d[i] = (dir[0] * c2w[4 * i] + dir[1] * c2w[4 * i + 1]) + dir[2] * c2w[4 * i + 2];
        o[i] = c2w[4 * i + 3];
    }
}
ORA_API void ora_get_rays(const float* directions, const float* poses, int per_ray_pose, int n, float* rays_o, float* rays_d) {
    for (int k = 0; k < n; ++k)
        ray_from_pose(poses + (per_ray_pose ? 12 * (size_t)k : 0), directions + 3 * (size_t)k, rays_o + 3 * (size_t)k, rays_d + 3 * (size_t)k);
}
/* training-batch sampling -- datasets/base.py:34-61 (rays[img_idxs, pix_idxs][:, :3], poses[img_idxs], directions[pix_idxs])
 * followed by get_rays (train.py:184) */
ORA_API void ora_sample_rays(const float* poses, const float* directions, const float* rays, int ray_c, long long hw,
                             const int64_t* img_idx, long long img0, const int64_t* pix_idx, int n, float* rays_o, float* rays_d,
                             float* rgb) {
    for (int k = 0; k < n; ++k) {
        const long long img = img_idx ? img_idx[k] : img0, pix = pix_idx[k];
        ray_from_pose(poses + 12 * img, directions + 3 * pix, rays_o + 3 * (size_t)k, rays_d + 3 * (size_t)k);
        if (rgb) for (int c = 0; c < 3; ++c) rgb[3 * (size_t)k + c] = rays[((size_t)img * hw + pix) * ray_c + c];
    }
}

/* a-10 grid utilities -- modules/utils.py:120-169 */
ORA_API void ora_morton3d(const int32_t* coords, int m, int32_t* indices) {
    for (int i = 0; i < m; ++i) indices[i] = (int32_t)morton3d((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}
ORA_API void ora_morton3d_invert(const int32_t* indices, int m, int32_t* coords) {
    for (int i = 0; i < m; ++i) {
        uint32_t ind = (uint32_t)indices[i];
        coords[3 * i] = morton3d_invert1(ind >> 0); coords[3 * i + 1] = morton3d_invert1(ind >> 1); coords[3 * i + 2] = morton3d_invert1(ind >> 2);
    }
}
ORA_API void ora_packbits(const float* grid, float thr, int n_bytes, uint8_t* out) {
    for (int n = 0; n < n_bytes; ++n) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; ++i) bits |= (grid[8 * (size_t)n + i] > thr) ? (uint8_t)(1u << i) : 0;
        out[n] = bits;
    }
}

/* ------------------------------------------------------------------------------------------------
 * f-1  distortion loss (next tier) -- modules/distortion.py:15-119, f32 literal
 * ---------------------------------------------------------------------------------------------- */
ORA_API void ora_distortion_fwd(const float* ws, const float* deltas, const float* ts, const int32_t* rays_a,
                                int n_rays, float* loss, float* ws_inc /*[S]*/, float* wts_inc /*[S]*/) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n_rays; ++i) {
        int ray_idx = rays_a[3 * i], start = rays_a[3 * i + 1], N = rays_a[3 * i + 2];
        float ws_temp = 0.f, wst_temp = 0.f, loss_temp = 0.f;
        for (int n = 0; n < N; ++n) {
            int idx = start + n;
            float ws_exc = ws_temp, wts_exc = wst_temp;                 /* :33-35 */
            float wts = ws[idx] * ts[idx];                              /* distortion.py:150 */
            ws_temp += ws[idx]; wst_temp += wts;                        /* :37-38 */
            ws_inc[idx] = ws_temp; wts_inc[idx] = wst_temp;             /* :41-42 */
            float l = 2.f * (wst_temp * ws_exc - ws_temp * wts_exc) + (1.f / 3.f) * ws[idx] * ws[idx] * deltas[idx]; /* :63 */
            loss_temp += l;                                             /* :82 */
        }
        loss[ray_idx] = loss_temp;                                      /* :84 */
    }
}
ORA_API void ora_distortion_bwd(const float* dL_dloss, const float* deltas, const float* ws, const float* ts,
                                const float* ws_inc, const float* wts_inc, const int32_t* rays_a, int n_rays, float* dL_dws) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n_rays; ++i) {
        int ray_idx = rays_a[3 * i], start = rays_a[3 * i + 1], N = rays_a[3 * i + 2];
        if (N <= 0) continue;
        int end_idx = start + N - 1;                                    /* :102 */
        float ws_sum = ws_inc[end_idx], wts_sum = wts_inc[end_idx];     /* :104-105 */
        for (int n = 0; n < N; ++n) {
            int idx = start + n;
            float selector = (idx == start) ? 0.f : ts[idx] * ws_inc[idx - 1] - wts_inc[idx - 1];   /* :112 */
            float g = dL_dloss[ray_idx] * 2.f * (selector + (wts_sum - wts_inc[idx] - ts[idx] * (ws_sum - ws_inc[idx]))); /* :114 */
            g += dL_dloss[ray_idx] * (2.f / 3.f) * ws[idx] * deltas[idx];                           /* :116 */
            dL_dws[idx] = g;                                            /* :118 */
        }
    }
}
