// mlp.hip -- the fused tiny-MLP (density + colour heads) of Instant-NGP on gfx950 matrix cores.
//
// Replaces, for the default architecture of the reference (modules/networks.py:111-132,136-166,293-380 under
// torch.autocast(fp16), train.py:177):
//     xyz_encoder : 32 -> 64 (ReLU) -> 16            sigma = exp(h[0])                (TruncExp, networks.py:18-30)
//     dir path    : d/|d| -> (d+1)/2 -> SH16                                          (networks.py:162-163)
//     rgb_net     : [SH16 | h16] -> 64 (ReLU) -> 64 (ReLU) -> 3 (Sigmoid)
// i.e. 5 forward + 10 backward hipBLASLt GEMMs and ~25 elementwise launches per step in the reference's formulation.
//
// Formulation: D[out_feature][sample] = W[out][k] * X[k][sample] with v_mfma_f32_16x16x32_f16, so that a wave owns
// 16-sample column tiles and the activations of one layer feed the next layer's B operand straight from the
// accumulator registers: lane (n = lane&15, g = lane>>4) of a D tile holds rows 4g..4g+3 of column n, and the B
// operand wants 8 k-values of column n per lane, so two D tiles (rows 4g+r of features [32s,32s+16) and
// [32s+16,32s+32)) ARE one K=32 B fragment if the weight fragment's k-order is permuted to match
// (chain(s,g,j) = 32s + 16(j>>2) + 4g + (j&3)).  The permutation is paid once, in the weight packer.
// Activations never touch LDS or HBM in the forward pass; fp16 rounding points are the ones autocast produces
// (every Linear output is fp16, accumulation fp32).
//
// Backward recomputes the forward per tile (cheaper than storing 0.68 KB/sample of activations), propagates
// dX = W^T dZ through the same register-chaining trick, and forms the weight gradients dW = dZ X^T with the sample
// index as K: that needs [feature][sample] fragments, so every wave stores its dZ / X rows untransposed into a blocked
// LDS image (13 KB per 32-sample group) and the dW waves read them back with gfx950's transposing LDS read
// (ds_read_b64_tr_b16).  The 40 dW output tiles are distributed over the 12 waves of a block (each wave sums ITS tiles
// over all the block's samples), accumulate in fp32 registers across the whole persistent loop and leave the block as
// line-coalesced global atomics -- no cross-wave reduction.  It runs over the live-sample list of the trainer
// (ngp_mlp_bwd_live) or over all samples.
// Both kernels are VALU-issue bound, not MFMA bound (16-sample tiles: 4 output values per lane and MFMA to convert,
// activate and re-pack): everything between two MFMAs uses the packed f16 instructions (profiles/microbench/
// r01_mlp_bwd_breakdown.txt).
//
// Numerics are tolerance-checked against an fp32 torch restatement and against torch's own autocast path
// (tests/test_gpu_mlp.py); bf16/fp16 MFMA is used because this is the one genuine dense contraction on the path.
#include "ngp_device.h"
#include <stdlib.h>

namespace ngp {

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define NGP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

constexpr int N_FWD_FRAGS = 20;
constexpr int N_ALL_FRAGS = 42;
// fragment ids inside the packed weight image
constexpr int F_W1 = 0;    // [mt]        4
constexpr int F_W2 = 4;    // [s]         2
constexpr int F_W3 = 6;    // [mt]        4
constexpr int F_W4 = 10;   // [mt*2+s]    8
constexpr int F_W5 = 18;   // [s]         2
constexpr int B_W5T = 20;  // [mt]        4
constexpr int B_W4T = 24;  // [mt*2+s]    8
constexpr int B_W3T = 32;  // [s]         2
constexpr int B_W2T = 34;  // [mt]        4
constexpr int B_W1T = 38;  // [mt*2+s]    4

__device__ __forceinline__ int chain_k(int s, int g, int j) { return 32 * s + 16 * (j >> 2) + 4 * g + (j & 3); }

// Pair-major encoding layout (hash_grid.hip, hash_fwd_f32_xcd_kernel): plane p holds [level p f0,f1 | level 15-p f0,f1].
// natural feature index (2*level + f) of slot `slot` in plane `p`:
__device__ __forceinline__ int pair_nat(int p, int slot) { return slot < 2 ? 2 * p + slot : 2 * (15 - p) + (slot - 2); }
// k-slot (g, j) of the layer-1 B fragment: natural layout reads features 8g..8g+7; pair layout reads planes 2g, 2g+1
__device__ __forceinline__ int enc_feat(int pairs, int g, int j) { return pairs ? pair_nat(2 * g + (j >> 2), j & 3) : 8 * g + j; }
// row i of M-tile mt of W1^T (= which enc feature a d_enc accumulator row is): pair layout makes rows 4g'..4g'+3 one plane
__device__ __forceinline__ int enc_row(int pairs, int mt, int i) { return pairs ? pair_nat(4 * mt + (i >> 2), i & 3) : 16 * mt + i; }

// One thread per (fragment, lane, slot): gathers the fp32 master weight, rounds to fp16 (what autocast's
// weight.to(fp16) does) and stores it where the MFMA A operand of that lane wants it.
__device__ __forceinline__ void pack_one(int tid, const float* W1, const float* W2, const float* W3, const float* W4,
                                         const float* W5, int pairs, half_t* wpack) {
    const int j = tid & 7, lane = (tid >> 3) & 63, frag = tid >> 9;
    const int i = lane & 15, g = lane >> 4;
    float v = 0.0f;
    if (frag < F_W2) {                                  // W1 [64][32], plain k
        const int mt = frag - F_W1;
        v = W1[(16 * mt + i) * 32 + enc_feat(pairs, g, j)];
    } else if (frag < F_W3) {                           // W2 [16][64], chained over a1
        const int s = frag - F_W2;
        v = W2[i * 64 + chain_k(s, g, j)];
    } else if (frag < F_W4) {                           // W3 [64][32], k = [SH(4g+j) | 16 + h(4g+j-4)]
        const int mt = frag - F_W3;
        const int k = (j < 4) ? (4 * g + j) : (16 + 4 * g + (j - 4));
        v = W3[(16 * mt + i) * 32 + k];
    } else if (frag < F_W5) {                           // W4 [64][64], chained over a3
        const int mt = (frag - F_W4) >> 1, s = (frag - F_W4) & 1;
        v = W4[(16 * mt + i) * 64 + chain_k(s, g, j)];
    } else if (frag < B_W5T) {                          // W5 [3][64] (rows >= 3 are padding), chained over a4
        const int s = frag - F_W5;
        v = (i < 3) ? W5[i * 64 + chain_k(s, g, j)] : 0.0f;
    } else if (frag < B_W4T) {                          // W5^T: rows = a4 features, k = dz5 (slot j<4 <-> 4g+j)
        const int mt = frag - B_W5T;
        const int k = 4 * g + j;
        v = (j < 4 && k < 3) ? W5[k * 64 + 16 * mt + i] : 0.0f;
    } else if (frag < B_W3T) {                          // W4^T: rows = a3 features, k = dz4 chained
        const int mt = (frag - B_W4T) >> 1, s = (frag - B_W4T) & 1;
        v = W4[chain_k(s, g, j) * 64 + 16 * mt + i];
    } else if (frag < B_W2T) {                          // W3^T restricted to the h inputs (16..31), k = dz3 chained
        const int s = frag - B_W3T;
        v = W3[chain_k(s, g, j) * 32 + 16 + i];
    } else if (frag < B_W1T) {                          // W2^T: rows = a1 features, k = dh (slot j<4 <-> 4g+j)
        const int mt = frag - B_W2T;
        v = (j < 4) ? W2[(4 * g + j) * 64 + 16 * mt + i] : 0.0f;
    } else {                                            // W1^T: rows = enc features, k = dz1 chained
        const int mt = (frag - B_W1T) >> 1, s = (frag - B_W1T) & 1;
        v = W1[chain_k(s, g, j) * 32 + enc_row(pairs, mt, i)];
    }
    wpack[tid] = (half_t)v;
}

__global__ void __launch_bounds__(256) mlp_pack_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                       const float* __restrict__ W3, const float* __restrict__ W4,
                                                       const float* __restrict__ W5, int pairs, half_t* __restrict__ wpack) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < N_ALL_FRAGS * 64 * 8) pack_one(tid, W1, W2, W3, W4, W5, pairs, wpack);
}

// Trainer fusion: Adam on the 9 408 flat MLP weights (same arithmetic as optim.hip's adam_kernel, state layout in
// include/ngp_hip.h) followed by the fp16 fragment repack for the next step, in ONE block -- replaces two launches.
// Adam on the 9408 MLP weights, then the fp16 fragment repack for the next step (one block: the repack reads what the block's
// own threads just wrote)
// Round 5: the block is the whole launch once the table's optimizer rides in the scatter-add's flush (hash_bwd_lds.hip), so its
// latency counts: every thread requests its ten weights' p / g / m / v at once (one round trip instead of ten dependent ones), the
// updated weights are also kept in LDS, and the repack reads them from there instead of from memory the block has just written.
__device__ __forceinline__ void adam_mlp_pack_block(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const float* __restrict__ sf,
                                                    const int32_t* __restrict__ si, float beta1, float beta2, float eps,
                                                    int pairs, half_t* __restrict__ wpack) {
    __shared__ float wl[9408];
    const bool skip = si[SI_SKIP] != 0;
    const float inv_scale = sf[SF_INV_SCALE], step_size = sf[SF_LR] / sf[SF_BC1], bc2_sqrt = sf[SF_BC2_SQRT];
    constexpr int PER = (9408 + 1023) / 1024;
    float pi[PER], gi[PER], mi[PER], vi[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = (int)threadIdx.x + 1024 * k, ic = i < 9408 ? i : 0;
        pi[k] = p[ic]; gi[k] = g[ic]; mi[k] = m[ic]; vi[k] = v[ic];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = (int)threadIdx.x + 1024 * k;
        if (i >= 9408) continue;
        if (!skip) {
            const float gr = gi[k] * inv_scale;
            const float mk = mi[k] + (gr - mi[k]) * (1.0f - beta1);
            const float vk = vi[k] * beta2 + gr * gr * (1.0f - beta2);
            const float denom = sqrtf(vk) / bc2_sqrt + eps;
            pi[k] = pi[k] - step_size * (mk / denom);
            p[i] = pi[k]; m[i] = mk; v[i] = vk;
        }
        g[i] = 0.0f;
        wl[i] = pi[k];
    }
    __syncthreads();
    // one 16-byte fragment slot (8 halfs) per work item
    for (int item = threadIdx.x; item < N_ALL_FRAGS * 64; item += blockDim.x) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pack_one(8 * item + j, wl, wl + 2048, wl + 3072, wl + 5120, wl + 9216, pairs, wpack);
    }
}

__global__ void __launch_bounds__(1024) adam_mlp_pack_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ sf,
                                                             const int32_t* __restrict__ si, float beta1, float beta2, float eps,
                                                             int pairs, half_t* __restrict__ wpack) {
    adam_mlp_pack_block(p, g, m, v, sf, si, beta1, beta2, eps, pairs, wpack);
}

// The whole optimizer pass in ONE launch: block 0 (dispatched first) does the MLP weights + repack while blocks 1.. stream the
// hash table -- the 16 us small-kernel tail of the step disappears under the 46 us table pass.
template <int SHADOW, bool GRAD16>
__global__ void __launch_bounds__(1024) adam_all_kernel(float4* __restrict__ tp, void* __restrict__ tg, float4* __restrict__ tm,
                                                        float4* __restrict__ tv, long n4, uint2* __restrict__ shadow,
                                                        float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ sf,
                                                        const int32_t* __restrict__ si, float beta1, float beta2, float eps,
                                                        int pairs, half_t* __restrict__ wpack) {
    if (blockIdx.x == 0) { if (p) adam_mlp_pack_block(p, g, m, v, sf, si, beta1, beta2, eps, pairs, wpack); }     // (mlp == NULL: table range only)
    else adam_table_pass<SHADOW, GRAD16>(tp, tg, tm, tv, n4, sf, si, beta1, beta2, eps, shadow, (long)blockIdx.x - 1, (long)gridDim.x - 1);
}

// ---- per-lane helpers -------------------------------------------------------------------------------------
// The data path is VALU-issue bound (ISA count: ~20 VALU per MFMA before this formulation), so everything between two MFMAs
// is written with the packed f16 instructions of gfx950: v_cvt_pk_f16_f32 (round-to-nearest-even, two values per issue),
// v_pk_max_f16 / v_pk_min_f16, v_pk_min_u16 / v_pk_mul_lo_u16 -- no per-element compare + select, no lane masks in SGPRs.
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4 to_h4(const floatx4& d) { return __builtin_convertvector(d, half4); }
__device__ __forceinline__ half4 relu_h4(const floatx4& d) {             // relu(f16(d)); max(-0, +0) = +0 on AMD
    const half4 z = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
    return __builtin_elementwise_max(to_h4(d), z);
}
__device__ __forceinline__ half8 cat_h4(const half4& a, const half4& b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// dz = (act > 0) ? f16(d) : 0   (threshold_backward).  act >= +0, so (0 - bits(act)) as int16 is negative exactly when
// act > 0: an arithmetic shift by 15 turns that into the 0xFFFF / 0 select mask, and one AND applies it -- an active unit keeps
// every bit of f16(d) (inf / NaN included, for the GradScaler check), a masked one gives +0.
__device__ __forceinline__ half4 mask_h4(const floatx4& d, const half4& act) {
    typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
    const uint2v a = __builtin_bit_cast(uint2v, act), h = __builtin_bit_cast(uint2v, to_h4(d));
    uint2v r;
    // written out: the C form becomes per-element compare + select
    asm("v_pk_sub_i16 %0, 0, %2\n\t"
        "v_pk_sub_i16 %1, 0, %3\n\t"
        "v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]\n\t"
        "v_pk_ashrrev_i16 %1, 15, %1 op_sel_hi:[0,1]\n\t"
        "v_and_b32 %0, %0, %4\n\t"
        "v_and_b32 %1, %1, %5"
        : "=&v"(r.x), "=&v"(r.y) : "v"(a.x), "v"(a.y), "v"(h.x), "v"(h.y));
    return __builtin_bit_cast(half4, r);
}

// 1-ulp hardware approximations (v_exp_f32 / v_rcp_f32 / v_rsq_f32) where the result is rounded to fp16 anyway: the
// IEEE-exact expf / division / sqrt sequences were ~100 of the ~550 VALU instructions of a backward round
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// SH coefficients 4g..4g+3 of the encoded direction (x,y,z) = (d/|d| + 1)/2, spherical_harmonics.py:27-42
__device__ __forceinline__ half4 sh_quad(int g, float x, float y, float z) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float a, b, c, d;
    if (g == 0) {
        a = 0.28209479177387814f; b = -0.48860251190291987f * y; c = 0.48860251190291987f * z; d = -0.48860251190291987f * x;
    } else if (g == 1) {
        a = 1.0925484305920792f * xy; b = -1.0925484305920792f * yz; c = 0.94617469575755997f * z2 - 0.31539156525251999f;
        d = -1.0925484305920792f * xz;
    } else if (g == 2) {
        a = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; b = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        c = 2.8906114426405538f * xy * z; d = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    } else {
        a = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); b = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        c = 1.4453057213202769f * z * (x2 - y2); d = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
    half4 r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// The same, branch-free (mlp_bwd_reg_kernel: one wave per SIMD, the iteration has to stay ONE basic block for the two tiles'
// instruction streams to interleave).  ~20 more live VGPRs than the branchy form: not for the forward kernel, which sits at 156.
__device__ __forceinline__ half4 sh_quad_flat(int g, float x, float y, float z) {
    // all 16 coefficients, then a three-deep select on g: the four lane groups of a wave need different quads, and a branch per
    // quad is four exec-masked regions every wave walks through anyway -- this way the caller stays one basic block
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float a0 = 0.28209479177387814f, b0 = -0.48860251190291987f * y, c0 = 0.48860251190291987f * z, d0 = -0.48860251190291987f * x;
    const float a1 = 1.0925484305920792f * xy, b1 = -1.0925484305920792f * yz, c1 = 0.94617469575755997f * z2 - 0.31539156525251999f,
                d1 = -1.0925484305920792f * xz;
    const float a2 = 0.54627421529603959f * x2 - 0.54627421529603959f * y2, b2 = 0.59004358992664352f * y * (-3.0f * x2 + y2),
                c2 = 2.8906114426405538f * xy * z, d2 = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    const float a3 = 0.3731763325901154f * z * (5.0f * z2 - 3.0f), b3 = 0.45704579946446572f * x * (1.0f - 5.0f * z2),
                c3 = 1.4453057213202769f * z * (x2 - y2), d3 = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    const bool g0 = g == 0, g1 = g == 1, g2 = g == 2;
    half4 r;
    r[0] = (half_t)(g0 ? a0 : (g1 ? a1 : (g2 ? a2 : a3)));
    r[1] = (half_t)(g0 ? b0 : (g1 ? b1 : (g2 ? b2 : b3)));
    r[2] = (half_t)(g0 ? c0 : (g1 ? c1 : (g2 ? c2 : c3)));
    r[3] = (half_t)(g0 ? d0 : (g1 ? d1 : (g2 ? d2 : d3)));
    return r;
}

// the two 16-byte pieces of enc a lane (sample smp, group g) feeds into layer 1
__device__ __forceinline__ void enc_ptrs(const float* __restrict__ enc, int pairs, size_t plane, int smp, int g, const float*& p0,
                                         const float*& p1) {
    if (pairs) { p0 = enc + ((size_t)(2 * g) * plane + smp) * 4; p1 = enc + ((size_t)(2 * g + 1) * plane + smp) * 4; }
    else { p0 = enc + (size_t)smp * 32 + 8 * g; p1 = p0 + 4; }
}

struct TileFwd {            // everything the backward needs from the recomputed forward of one 16-sample tile
    half8 b_enc;            // layer-1 B operand (enc features 8g..8g+7)
    half4 a1[4];            // relu(L1), D layout (feature 16mt+4g+r)
    half4 h;                // L2 output (feature 4g+r)
    half8 b_in3;            // [SH(4g..4g+3) | h(4g..4g+3)]
    half4 a3[4], a4[4];
    half4 rgb;              // sigmoid(L5) rows 4g+r (only g==0, r<3 meaningful)
    float sigma;            // exp(h0), valid on g==0 lanes
};

__device__ __forceinline__ const half8& wfrag(const half8* __restrict__ wl, int id, int lane) { return wl[id * 64 + lane]; }

// forward of one 16-sample tile from registers; wl = packed weight image in LDS
// W(id) = this lane's piece of weight fragment `id` (from the LDS image, or from registers: mlp_bwd_reg_kernel)
template <bool COLOR, bool FLAT_SH = false, class WF>
__device__ __forceinline__ void tile_forward_frags(const WF& W, int g, const float4& e0, const float4& e1,
                                                   float dx, float dy, float dz, TileFwd& t) {
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    t.b_enc[0] = (half_t)e0.x; t.b_enc[1] = (half_t)e0.y; t.b_enc[2] = (half_t)e0.z; t.b_enc[3] = (half_t)e0.w;
    t.b_enc[4] = (half_t)e1.x; t.b_enc[5] = (half_t)e1.y; t.b_enc[6] = (half_t)e1.z; t.b_enc[7] = (half_t)e1.w;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) t.a1[mt] = relu_h4(NGP_MFMA(W(F_W1 + mt), t.b_enc, zero));
    floatx4 d2 = NGP_MFMA(W(F_W2 + 0), cat_h4(t.a1[0], t.a1[1]), zero);
    d2 = NGP_MFMA(W(F_W2 + 1), cat_h4(t.a1[2], t.a1[3]), d2);
    t.h = to_h4(d2);
    t.sigma = expf((float)t.h[0]);
    if (!COLOR) return;
    const float inv = fast_rsq(dx * dx + dy * dy + dz * dz);
    const float x = (dx * inv + 1.0f) / 2.0f, y = (dy * inv + 1.0f) / 2.0f, z = (dz * inv + 1.0f) / 2.0f;
    t.b_in3 = cat_h4(FLAT_SH ? sh_quad_flat(g, x, y, z) : sh_quad(g, x, y, z), t.h);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) t.a3[mt] = relu_h4(NGP_MFMA(W(F_W3 + mt), t.b_in3, zero));
    const half8 b30 = cat_h4(t.a3[0], t.a3[1]), b31 = cat_h4(t.a3[2], t.a3[3]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        floatx4 d4 = NGP_MFMA(W(F_W4 + 2 * mt), b30, zero);
        d4 = NGP_MFMA(W(F_W4 + 2 * mt + 1), b31, d4);
        t.a4[mt] = relu_h4(d4);
    }
    floatx4 d5 = NGP_MFMA(W(F_W5 + 0), cat_h4(t.a4[0], t.a4[1]), zero);
    d5 = NGP_MFMA(W(F_W5 + 1), cat_h4(t.a4[2], t.a4[3]), d5);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float c = (float)(half_t)d5[r];
        t.rgb[r] = (half_t)fast_rcp(1.0f + fast_exp(-c));
    }
}

template <bool COLOR>
__device__ __forceinline__ void tile_forward_regs(const half8* __restrict__ wl, int lane, int g, const float4& e0, const float4& e1,
                                                  float dx, float dy, float dz, TileFwd& t) {
    tile_forward_frags<COLOR>([&](int id) -> const half8& { return wl[id * 64 + lane]; }, g, e0, e1, dx, dy, dz, t);
}

// the packed weight image -> LDS.  All of a thread's 16-byte pieces are requested before the first one is stored (the plain loop
// waited for every load before issuing the next: 5 serial L2 round trips at the head of each of the forward's 768 blocks)
template <int THREADS, int N_FRAGS>
__device__ __forceinline__ void load_wpack(const half_t* __restrict__ wpack, half8* __restrict__ wl) {
    const uint4* src = reinterpret_cast<const uint4*>(wpack);
    uint4* dst = reinterpret_cast<uint4*>(wl);
    constexpr int PER = (N_FRAGS * 64 + THREADS - 1) / THREADS;
    uint4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = threadIdx.x + i * THREADS;
        v[i] = src[k < N_FRAGS * 64 ? k : 0];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = threadIdx.x + i * THREADS;
        if (k < N_FRAGS * 64) dst[k] = v[i];
    }
    __syncthreads();
}

// ---- forward kernel: persistent waves, 32 samples (two 16-sample tiles) per trip; the next trip's inputs are requested
//      before this trip's math (the loads are the only long-latency operations of a trip).  Round 4: the request is branch-free
//      (a lane past the end reads sample 0; zeros are SELECTED where the values are consumed, one trip later) -- with the loads
//      inside `if (smp < S)` the compiler merged them with the defaults at the end of that block and waited for them right there
//      (s_waitcnt vmcnt(2) three instructions behind the "prefetch"): 22 -> ... us at 400 k samples.  PAIRS is a template
//      parameter so that the uniform branch on the encoding layout does not split the request into blocks either. ------------
struct FwdIn { float4 e0, e1; float dx, dy, dz; };
template <bool COLOR, bool PAIRS>
__device__ __forceinline__ void fwd_request(FwdIn& raw, const float* __restrict__ enc, const float* __restrict__ dirs, int smp, int S,
                                            int g, size_t plane) {
    const size_t s0 = smp < S ? (size_t)smp : 0;
    const float *ep0, *ep1;
    enc_ptrs(enc, PAIRS ? 1 : 0, plane, (int)s0, g, ep0, ep1);
    raw.e0 = *reinterpret_cast<const float4*>(ep0);
    raw.e1 = *reinterpret_cast<const float4*>(ep1);
    if (COLOR) { raw.dx = dirs[3 * s0]; raw.dy = dirs[3 * s0 + 1]; raw.dz = dirs[3 * s0 + 2]; }
    else { raw.dx = 0.f; raw.dy = 0.f; raw.dz = 1.f; }
}
__device__ __forceinline__ FwdIn fwd_take(const FwdIn& raw, bool ok) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    FwdIn in;
    in.e0 = ok ? raw.e0 : z4; in.e1 = ok ? raw.e1 : z4;
    in.dx = ok ? raw.dx : 0.f; in.dy = ok ? raw.dy : 0.f; in.dz = ok ? raw.dz : 1.f;
    return in;
}

// LIST (round 5, the chunked forward): trip positions index a list of samples -- position j shades sample list[j] (reads its enc
// row and direction, writes its sigma / rgb) -- instead of the samples themselves; the list entries of the trip after next are
// requested with the next trip's inputs, so the indirection stays off the trip's critical path.
template <bool COLOR, bool PAIRS, bool LIST = false>
__global__ void __launch_bounds__(256) mlp_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ dirs,
                                                      const half_t* __restrict__ wpack, int S,
                                                      const int32_t* __restrict__ n_dev, float* __restrict__ sigmas,
                                                      half_t* __restrict__ rgbs, const int32_t* __restrict__ list = nullptr) {
    __shared__ half8 wl[N_FWD_FRAGS * 64];
    const size_t plane = (size_t)S;
    if (n_dev) S = min(S, *n_dev);
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    const int n_iter = (S + 31) >> 5;
    if (LIST && S <= 0) return;          // (an empty list: entry 0 is not a list entry -- nothing may be dereferenced through it)
    // LIST: a position past the end reads entry 0 of the list, which exists
    auto entry = [&](int pos) -> int { return LIST ? list[(pos < S) ? pos : 0] : pos; };
    FwdIn nxt[2];
    int s_cur[2] = {0, 0}, s_nxt[2], s_nxt2[2] = {0, 0};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        s_nxt[tt] = entry(wave * 32 + 16 * tt + n);
        if (LIST) s_nxt2[tt] = entry((wave + n_waves) * 32 + 16 * tt + n);
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        if (LIST) fwd_request<COLOR, PAIRS>(nxt[tt], enc, dirs, (wave * 32 + 16 * tt + n < S) ? s_nxt[tt] : 0, (int)plane, g, plane);
        else fwd_request<COLOR, PAIRS>(nxt[tt], enc, dirs, wave * 32 + 16 * tt + n, S, g, plane);
    }
    load_wpack<256, COLOR ? N_FWD_FRAGS : F_W3>(wpack, wl);          // (behind the first trip's input request: both in flight together)
    for (int it = wave; it < n_iter; it += n_waves) {
        FwdIn cur[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) { cur[tt] = fwd_take(nxt[tt], it * 32 + 16 * tt + n < S); s_cur[tt] = s_nxt[tt]; }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pos = (it + n_waves) * 32 + 16 * tt + n;
            if (LIST) {
                s_nxt[tt] = s_nxt2[tt];
                s_nxt2[tt] = entry((it + 2 * n_waves) * 32 + 16 * tt + n);
                fwd_request<COLOR, PAIRS>(nxt[tt], enc, dirs, (pos < S) ? s_nxt[tt] : 0, (int)plane, g, plane);
            } else {
                fwd_request<COLOR, PAIRS>(nxt[tt], enc, dirs, pos, S, g, plane);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pos = it * 32 + 16 * tt + n;
            const int smp = LIST ? s_cur[tt] : pos;
            TileFwd t;
            tile_forward_regs<COLOR>(wl, lane, g, cur[tt].e0, cur[tt].e1, cur[tt].dx, cur[tt].dy, cur[tt].dz, t);
            if (pos < S && g == 0) {
                sigmas[smp] = t.sigma;
                if (COLOR) { rgbs[3 * (size_t)smp] = t.rgb[0]; rgbs[3 * (size_t)smp + 1] = t.rgb[1]; rgbs[3 * (size_t)smp + 2] = t.rgb[2]; }
            }
        }
    }
}

// ---- backward kernel ------------------------------------------------------------------------------------------
// Weight gradients contract over SAMPLES, while the data path keeps one sample per lane: the operands have to be transposed
// through LDS.  Every wave stores its activations / pre-activation gradients UNtransposed, [sample row][feature column], with
// 8- and 16-byte stores straight from the MFMA register layout, and the dW waves read them back with gfx950's transposing LDS
// read (ds_read_b64_tr_b16): within a 16-lane group, lane i supplies the address of 4 contiguous halfs (sample row i/4,
// features 4(i%4)..) and lane c receives feature c of the group's 4 sample rows -- the A/B fragment of a K = samples MFMA.
// (Measured on the previous scheme, which transposed on the write side with 120 ds_write_b16 per lane and round: those
// stores were 31 % of the kernel, profiles/microbench/r01_mlp_bwd_breakdown.txt.)
// Image of one 32-sample group: [feature tile T][row block rb = sample row / 4][4 rows][16 features] halfs, i.e. every
// [4 samples][16 features] block a 16-lane tr-read group consumes is 128 contiguous bytes and the two blocks of a 32-lane LDS
// service group are adjacent (256 B = all 64 banks once): the layout ds_read_b64_tr_b16 reads without bank conflicts
// (row-strided images cost ~5 extra LDS cycles per read, SQ_LDS_BANK_CONFLICT; padding and XOR swizzles do not help there).
// Stores: a 16-lane ds_write_b64 group holds 16 sample rows of ONE granule (4 halfs) column, which would land 4-way on the
// same banks; granule gi of a block is therefore kept at slot gi ^ (rb & 3) -- the group then covers all 32 banks once.
constexpr int IMG_TILE = 8 * 64;          // halfs per feature tile (8 row blocks x 128 B)
constexpr int IMG_TILES = 13;             // up to 208 feature columns are live at a time
constexpr int IMG_HALFS = IMG_TILES * IMG_TILE;     // 13 KB per group
// tile map of the three dW phases
constexpr int A_DZ5 = 0, A_DZ4 = 1, A_A4 = 5, A_A3 = 9;            // phase A: layers 5 and 4
constexpr int B_DZ3 = 0, B_DZ2 = 4, B_IN3 = 5, B_A1 = 7;           // phase B: layers 3 and 2
constexpr int C_DZ1 = 0, C_ENC = 4;                                // phase C: layer 1
constexpr int N_W = 2048 + 1024 + 2048 + 4096 + 192;    // 9408 weights
static_assert(N_W == NGP_MLP_NW, "ngp_device.h");
constexpr int OFF_W1 = 0, OFF_W2 = 2048, OFF_W3 = 3072, OFF_W4 = 5120, OFF_W5 = 9216;

typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;
// lofs = this lane's (row block, row, swizzled granule) offset inside a tile
__device__ __forceinline__ void img_store(half_t* img, int tile, int lofs, const half4& v) {
    *reinterpret_cast<half4*>(img + tile * IMG_TILE + lofs) = v;
}
// fragment of feature tile `tile` over the group's 32 samples: k-slot (q, j) = sample row 4q + j (j < 4) or 16 + 4q + (j - 4)
// (row blocks q and q + 4, same swizzle key); both operands of an MFMA use the same mapping, which is all a contraction needs
__device__ __forceinline__ half8 img_load_tr(const half_t* img, int tile, int q, int i) {
    const half_t* p = img + tile * IMG_TILE + q * 64 + (i >> 2) * 16 + 4 * ((i & 3) ^ q);
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p + 4 * 64));
    return cat_h4(__builtin_bit_cast(half4, lo), __builtin_bit_cast(half4, hi));
}

// Block = 12 waves (3 per SIMD).  Every wave runs the data path (forward recompute + dX chain) of ONE 16-sample tile; waves
// 2k and 2k+1 form the 32-sample group k (image rows 16 * (wave & 1) + n).  The 40 dW output tiles (16x16 each) are
// DISTRIBUTED over the 12 waves (4 accumulators per wave instead of 160 in a wave-private scheme); each wave accumulates its
// tiles over the images of all 6 groups (K = 192 samples per round).  Three store -> barrier -> accumulate -> barrier phases
// per round, two layers at a time so that both wave classes have work in a phase:
//     phase A: layer 4 (waves 0..7, two tiles each)  + layer 5 (waves 8..11)
//     phase B: layer 3 (waves 0..7)                  + layer 2 (waves 8..11)
//     phase C: layer 1 (waves 0..7)
// A block's waves own disjoint dW tiles, so nothing is reduced across waves at the end.
// Round 3 (profiles/r03_mlp_bwd_experiments.txt): at 350-400 k samples a round of 192 samples takes ~8.3 us, 62 % of it in the
// three dW phases, with the LDS 27 %, the VALU 25 % and the MFMA pipe 14 % busy -- a phase uses one resource at a time and six
// block-wide barriers per round keep all 12 waves in the same phase.  Two independent 6-wave blocks per CU (77 KB of LDS each)
// would interleave, but do not become co-resident: a workgroup's waves go to the SIMDs in cyclic order (2,2,1,1), and at
// 164 VGPRs (3 waves per SIMD) a second such workgroup finds no complementary slots (profiles/microbench/occupancy_probe.hip:
// 384-thread blocks pair up per CU at <= 128 VGPRs only) -- measured 98 us instead of 75.  8-wave blocks would need 86 KB.
constexpr int BW = 12;                     // waves per block
constexpr int BG = BW / 2;                 // 32-sample groups per round

struct BwdIn {                             // prefetched per-round inputs of one lane
    float4 e0, e1;                         // enc features 8g..8g+7
    float dx, dy, dz, dsig;
    uint32_t rg, b;                        // the three fp16 colour gradients as loaded (unpacked at the use: packing them at the load
                                           // puts a VALU -- and with it an s_waitcnt vmcnt(0) -- straight behind the prefetch)
    __device__ __forceinline__ float drgb(int r) const {
        const uint16_t bits = r == 0 ? (uint16_t)(rg & 0xffffu) : (r == 1 ? (uint16_t)(rg >> 16) : (uint16_t)b);
        return (float)__builtin_bit_cast(half_t, bits);
    }
};

// position smp of the (optionally compacted) backward list -> sample index, -1 past the end
__device__ __forceinline__ int bwd_src(int smp, int S, const int32_t* __restrict__ idx) {
    const int q = smp < S ? smp : 0;                       // (branch-free: entry 0 always exists, the launch has n_max >= 1)
    const int v = idx ? idx[q] : q;
    return smp < S ? v : -1;
}
// the same in two halves for the register kernel: the LOAD of the entry (entry 0 past the end) and, where its value is consumed,
// the select that turns a position past the end into -1
template <bool LIST>
__device__ __forceinline__ int bwd_entry(int smp, int S, const int32_t* __restrict__ idx) {
    const int q = smp < S ? smp : 0;
    return LIST ? idx[q] : q;
}
// Branch-free, and split in two: bwd_request issues the loads of a lane's inputs (a lane past the end reads sample 0) and touches
// none of the results; bwd_take, called where the data path starts to consume them, selects zeros for lanes past the end
// (dir = +z) and a zero gradient for lanes g != 0.  A select behind the loads would sit in the requesting round and make it end
// with s_waitcnt vmcnt(0) -- draining the round's own d_enc stores too.
struct BwdRaw {
    float4 e0, e1;
    float dx, dy, dz, ds;
    uint32_t rg, b;
    int src;
};
__device__ __forceinline__ void bwd_request(BwdRaw& raw, const float* __restrict__ enc, const float* __restrict__ dirs,
                                            const float* __restrict__ dsigmas, const half_t* __restrict__ drgbs, int src,
                                            int g, int pairs, size_t plane) {
    const size_t s0 = src >= 0 ? (size_t)src : 0;
    const float *ep0, *ep1;
    enc_ptrs(enc, pairs, plane, (int)s0, g, ep0, ep1);
    raw.e0 = *reinterpret_cast<const float4*>(ep0); raw.e1 = *reinterpret_cast<const float4*>(ep1);
    raw.dx = dirs[3 * s0]; raw.dy = dirs[3 * s0 + 1]; raw.dz = dirs[3 * s0 + 2]; raw.ds = dsigmas[s0];
    // the row's three halfs as two overlapping dwords [r g] [g b]: a 16-bit load writes half a register, and the copy that merges it
    // into the loop-carried one is a VALU the scheduler parks in the middle of the round, behind an s_waitcnt vmcnt(0)
    __builtin_memcpy(&raw.rg, drgbs + 3 * s0, 4);
    __builtin_memcpy(&raw.b, drgbs + 3 * s0 + 1, 4);
    raw.src = src;
}
__device__ __forceinline__ void bwd_take(BwdIn& in, const BwdRaw& raw, int g) {
    const bool ok = raw.src >= 0, head = ok && g == 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    in.e0 = ok ? raw.e0 : z4; in.e1 = ok ? raw.e1 : z4;
    in.dx = ok ? raw.dx : 0.f; in.dy = ok ? raw.dy : 0.f; in.dz = ok ? raw.dz : 1.f;
    in.dsig = head ? raw.ds : 0.f; in.rg = head ? raw.rg : 0u; in.b = head ? raw.b >> 16 : 0u;
}

#ifdef NGP_MLP_DIAG
// Timing builds only (-DNGP_MLP_DIAG, profiles/microbench/mlp_time.py): s_memtime ticks per segment of a round, summed over the
// rounds of block 0, per wave.  The counter update is a global read-modify-write, i.e. every stamp also drains the wave's
// outstanding global loads / stores: the prefetch below shows up in the segment that follows it.
__device__ unsigned long long ngp_mlp_dbg[16 * 8];
#define MLP_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && lane == 0) ngp_mlp_dbg[wv * 8 + (k)] += now_ - t_prev; t_prev = now_; } while (0)
#else
#define MLP_T(k) do { } while (0)
#endif

__global__ void __launch_bounds__(768) mlp_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ dirs,
                                                       const half_t* __restrict__ wpack, const float* __restrict__ dsigmas,
                                                       const half_t* __restrict__ drgbs, int S,
                                                       const int32_t* __restrict__ n_dev, const int32_t* __restrict__ idx, int pairs,
                                                       float* __restrict__ d_enc,
                                                       float* __restrict__ dW /*[N_W], pre-zeroed or accumulating*/,
                                                       float* __restrict__ parts /*[gridDim.x][N_W] or NULL*/,
                                                       int32_t* __restrict__ found_inf) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[N_ALL_FRAGS * 64 * 16 + BG * IMG_HALFS * 2];
    const size_t plane = (size_t)S;
    if (n_dev) S = min(S, *n_dev);
    half8* wl = reinterpret_cast<half8*>(smem);
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int grp = wv >> 1, par = wv & 1;
    half_t* Iall = reinterpret_cast<half_t*>(smem + N_ALL_FRAGS * 64 * 16);
    half_t* img = Iall + grp * IMG_HALFS;                                           // this wave pair's group image
    const int col = 16 * par + n;                                                   // sample slot inside the 32-sample group
    const int n_iter = (S + 31) >> 5;
    const int n_round = (n_iter + BG - 1) / BG;
    const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
    const half4 hzero = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};

    // dW tile ownership (40 tiles of 16x16 over 12 waves):
    //   waves 0..7  (w)    : L4 tiles (mt = w>>1, nt = 2(w&1), 2(w&1)+1) -> accA0/accA1 ;  L3 tile (mt = w>>1, nt = w&1) -> accB ;
    //                        L1 tile (mt = w>>1, nt = w&1) -> accC
    //   waves 8..11 (v=w-8): L2 tile (nt = v) -> accB ;  L5 tile (nt = v) -> accC
    floatx4 accA0 = zero, accA1 = zero, accB = zero, accC = zero;
    bool bad_denc = false;             // a non-finite d_enc value: the same condition the scatter-add flags on its input
    const bool lo8 = wv < 8;
    const int v4 = wv - 8;
    const int mtL = wv >> 1, ntL = wv & 1;         // lo8: tile row / column of the L3 and L1 tiles, tile row of the L4 pair
    // image row 16 par + n -> row block 4 par + (n >> 2), row n & 3, swizzle key n >> 2; this lane's D-layout granule is g
    const int lrow = (4 * par + (n >> 2)) * 64 + (n & 3) * 16;
    const int lofs = lrow + 4 * (g ^ (n >> 2));

    BwdRaw raw;
    bwd_request(raw, enc, dirs, dsigmas, drgbs, bwd_src((blockIdx.x * BG + grp) * 32 + col, S, idx), g, pairs, plane);
    load_wpack<64 * BW, N_ALL_FRAGS>(wpack, wl);                     // (behind the first round's input request: both in flight together)
    for (int round = blockIdx.x; round < n_round; round += gridDim.x) {
        BwdIn in;
        bwd_take(in, raw, g);
        const int smp = (round * BG + grp) * 32 + col;
        // The next round's list entry is requested now and its inputs once this round's data path has consumed `in` (below):
        // the dependent idx -> enc load chain then runs underneath the three dW phases instead of in front of the next round
        // (74.5 -> 72.5 us at 400 k live samples, +10 VGPRs; with the uncompacted list of rounds 1-2 it measured no gain)
        int src_next = -1;
        if (round + (int)gridDim.x < n_round) src_next = bwd_src(((round + (int)gridDim.x) * BG + grp) * 32 + col, S, idx);
        TileFwd t;
        half4 dz5 = hzero, dz4[4], dz3[4], dz2, dz1[4];
#ifdef NGP_MLP_DIAG
        unsigned long long t_prev = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        MLP_T(0);
#endif
        tile_forward_regs<true>(wl, lane, g, in.e0, in.e1, in.dx, in.dy, in.dz, t);
        MLP_T(1);
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float y = (float)t.rgb[r];
                dz5[r] = (half_t)(in.drgb(r) * ((1.0f - y) * y));                               // sigmoid_backward
            }
        }
        {
            const half8 b_dz5 = cat_h4(dz5, hzero);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dz4[mt] = mask_h4(NGP_MFMA(wfrag(wl, B_W5T + mt, lane), b_dz5, zero), t.a4[mt]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 b40 = cat_h4(dz4[0], dz4[1]), b41 = cat_h4(dz4[2], dz4[3]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                floatx4 d = NGP_MFMA(wfrag(wl, B_W4T + 2 * mt, lane), b40, zero);
                d = NGP_MFMA(wfrag(wl, B_W4T + 2 * mt + 1, lane), b41, d);
                dz3[mt] = mask_h4(d, t.a3[mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
            floatx4 dh = NGP_MFMA(wfrag(wl, B_W3T + 0, lane), cat_h4(dz3[0], dz3[1]), zero);
            dh = NGP_MFMA(wfrag(wl, B_W3T + 1, lane), cat_h4(dz3[2], dz3[3]), dh);
            dz2 = to_h4(dh);
            if (g == 0) {                                                     // TruncExp backward, networks.py:28-30
                const float h0 = (float)t.h[0];
                const half_t gs = (half_t)(in.dsig * fast_exp(fminf(fmaxf(h0, -15.0f), 15.0f)));
                dz2[0] = (half_t)((float)dz2[0] + (float)gs);
            }
            __builtin_amdgcn_sched_barrier(0);
            const half8 b_dz2 = cat_h4(dz2, hzero);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dz1[mt] = mask_h4(NGP_MFMA(wfrag(wl, B_W2T + mt, lane), b_dz2, zero), t.a1[mt]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 b10 = cat_h4(dz1[0], dz1[1]), b11 = cat_h4(dz1[2], dz1[3]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                floatx4 d = NGP_MFMA(wfrag(wl, B_W1T + 2 * mt, lane), b10, zero);
                d = NGP_MFMA(wfrag(wl, B_W1T + 2 * mt + 1, lane), b11, d);
                bad_denc |= !(isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]) && isfinite(d[3]));
                if (smp < S) {       // D rows 4g..4g+3 of tile mt = natural features 16mt+4g.. or, pair layout, plane 4mt+g
                    float* dp = pairs ? d_enc + ((size_t)(4 * mt + g) * plane + smp) * 4 : d_enc + (size_t)smp * 32 + 16 * mt + 4 * g;
                    *reinterpret_cast<float4*>(dp) = make_float4(d[0], d[1], d[2], d[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        MLP_T(2);
        bwd_request(raw, enc, dirs, dsigmas, drgbs, src_next, g, pairs, plane);
        // ---- weight gradients: every wave publishes its dZ / X rows, then accumulates ITS dW tiles over all BG group images ----
        // phase A: layer 5 (dZ5 [16] x a4 [64]) and layer 4 (dZ4 [64] x a3 [64])
        img_store(img, A_DZ5, lofs, dz5);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            img_store(img, A_DZ4 + mt, lofs, dz4[mt]);
            img_store(img, A_A4 + mt, lofs, t.a4[mt]);
            img_store(img, A_A3 + mt, lofs, t.a3[mt]);
        }
        __syncthreads();
        if (lo8) {
#pragma unroll 3
            for (int k = 0; k < BG; ++k) {
                const half_t* Ik = Iall + k * IMG_HALFS;
                const half8 a = img_load_tr(Ik, A_DZ4 + mtL, g, n);
                accA0 = NGP_MFMA(a, img_load_tr(Ik, A_A3 + 2 * ntL, g, n), accA0);
                accA1 = NGP_MFMA(a, img_load_tr(Ik, A_A3 + 2 * ntL + 1, g, n), accA1);
            }
        } else {
#pragma unroll 3
            for (int k = 0; k < BG; ++k) {
                const half_t* Ik = Iall + k * IMG_HALFS;
                accC = NGP_MFMA(img_load_tr(Ik, A_DZ5, g, n), img_load_tr(Ik, A_A4 + v4, g, n), accC);
            }
        }
        __syncthreads();
        MLP_T(3);
        // phase B: layer 3 (dZ3 [64] x in3 [32], in3 = [SH 0..15 | h 0..15]) and layer 2 (dZ2 [16] x a1 [64])
        img_store(img, B_DZ2, lofs, dz2);
        img_store(img, B_IN3, lofs, __builtin_shufflevector(t.b_in3, t.b_in3, 0, 1, 2, 3));
        img_store(img, B_IN3 + 1, lofs, __builtin_shufflevector(t.b_in3, t.b_in3, 4, 5, 6, 7));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            img_store(img, B_DZ3 + mt, lofs, dz3[mt]);
            img_store(img, B_A1 + mt, lofs, t.a1[mt]);
        }
        __syncthreads();
        if (lo8) {
#pragma unroll 3
            for (int k = 0; k < BG; ++k) {
                const half_t* Ik = Iall + k * IMG_HALFS;
                accB = NGP_MFMA(img_load_tr(Ik, B_DZ3 + mtL, g, n), img_load_tr(Ik, B_IN3 + ntL, g, n), accB);
            }
        } else {
#pragma unroll 3
            for (int k = 0; k < BG; ++k) {
                const half_t* Ik = Iall + k * IMG_HALFS;
                accB = NGP_MFMA(img_load_tr(Ik, B_DZ2, g, n), img_load_tr(Ik, B_A1 + v4, g, n), accB);
            }
        }
        __syncthreads();
        MLP_T(4);
        // phase C: layer 1 (dZ1 [64] x enc [32]); X column c = k-slot (g' = c>>3, j' = c&7) of layer 1: this lane's 8g..8g+7
        // are granules 2(g&1), 2(g&1)+1 of tile g>>1
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) img_store(img, C_DZ1 + mt, lofs, dz1[mt]);
        {
            const int te = C_ENC + (g >> 1), sw = n >> 2, gi = 2 * (g & 1);
            img_store(img, te, lrow + 4 * (gi ^ sw), __builtin_shufflevector(t.b_enc, t.b_enc, 0, 1, 2, 3));
            img_store(img, te, lrow + 4 * ((gi + 1) ^ sw), __builtin_shufflevector(t.b_enc, t.b_enc, 4, 5, 6, 7));
        }
        __syncthreads();
        if (lo8) {
#pragma unroll 3
            for (int k = 0; k < BG; ++k) {
                const half_t* Ik = Iall + k * IMG_HALFS;
                accC = NGP_MFMA(img_load_tr(Ik, C_DZ1 + mtL, g, n), img_load_tr(Ik, C_ENC + ntL, g, n), accC);
            }
        }
        __syncthreads();
        MLP_T(5);
    }

    if (!dW && !parts) { if (found_inf && bad_denc) *found_inf = 1; return; }      // (weight gradients not wanted: d_enc only)
    // ---- each wave owns its dW tiles outright.  parts != NULL: the block's sums leave as plain stores into ITS slab (every weight
    // is owned by exactly one lane of the block, so the slab is complete) and a later launch adds the slabs up (mlp_dw_reduce_block);
    // otherwise line-coalesced global atomics straight from the accumulators: 256 blocks x 9408 float atomics on the same 9408
    // addresses, 13 of the launch's 72 us at 400 k samples (profiles/r04_mlp_bwd_experiments.txt) ----
    // D layout of a dW tile: row (out) = 16 mt + 4g + r, col (in) = 16 nt + n
    bool bad = false;
    float* slab = parts ? parts + (size_t)blockIdx.x * N_W : nullptr;
    auto leave = [&](int w, float v) {
        if (slab) slab[w] = v;
        else if (v != 0.0f) unsafeAtomicAdd(dW + w, v);
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * mtL + 4 * g + r, o2 = 4 * g + r;
        const float a0 = accA0[r], a1 = accA1[r], b = accB[r], c = accC[r];
        if (lo8) {
            leave(OFF_W4 + o * 64 + 32 * ntL + n, a0);
            leave(OFF_W4 + o * 64 + 32 * ntL + 16 + n, a1);
            leave(OFF_W3 + o * 32 + 16 * ntL + n, b);
            // column c of a dW1 tile is the c-th enc column = k-slot (g' = c>>3, j' = c&7) of layer 1
            leave(OFF_W1 + o * 32 + enc_feat(pairs, (16 * ntL + n) >> 3, n & 7), c);
        } else {
            leave(OFF_W2 + o2 * 64 + 16 * v4 + n, b);
            if (o2 < 3) leave(OFF_W5 + o2 * 64 + 16 * v4 + n, c);
        }
        bad |= !isfinite(a0) || !isfinite(a1) || !isfinite(b) || !isfinite(c);
    }
    if (found_inf && (bad || bad_denc)) *found_inf = 1;
}

#ifdef NGP_MLP_BWD_REG          // round 4's register-resident form: a recorded negative, built on request only (see the file's header)
#include "../../profiles/microbench/mlp_bwd_reg_kernel.inc"
#endif

}  // namespace ngp

using namespace ngp;

extern "C" {

#ifdef NGP_MLP_DIAG
int ngp_mlp_debug_read(unsigned long long* host128, int reset) {
    if (hipMemcpyFromSymbol(host128, HIP_SYMBOL(ngp_mlp_dbg), sizeof(ngp_mlp_dbg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16 * 8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(ngp_mlp_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

int ngp_mlp_wpack_halfs(void) { return N_ALL_FRAGS * 64 * 8; }

int ngp_mlp_pack(const float* W1, const float* W2, const float* W3, const float* W4, const float* W5, int enc_pairs, uint16_t* wpack,
                 void* stream) {
    const int total = N_ALL_FRAGS * 64 * 8;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W1, W2, W3, W4, W5, enc_pairs,
                       (half_t*)wpack);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_adam_mlp_pack(float* p, float* g, float* m, float* v, const float* state_f, const int32_t* state_i, float beta1, float beta2,
                      float eps, int enc_pairs, uint16_t* wpack, void* stream) {
    hipLaunchKernelGGL(adam_mlp_pack_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, g, m, v, state_f, state_i, beta1, beta2,
                       eps, enc_pairs, (half_t*)wpack);
    NGP_LAUNCH_CHECK();
    return 0;
}

#define NGP_ADAM_ALL(SH, G16)                                                                                                     \
    hipLaunchKernelGGL((adam_all_kernel<SH, G16>), grid, block, 0, (hipStream_t)stream, (float4*)table, table_g, (float4*)table_m,    \
                       (float4*)table_v, n4, (uint2*)table_16, mlp, mlp_g, mlp_m, mlp_v, state_f, state_i, beta1, beta2, eps,         \
                       enc_pairs, (half_t*)wpack)

int ngp_adam_all_ex(float* table, void* table_g, int grad_is_f16, float* table_m, float* table_v, long long n, uint16_t* table_16,
                    int copy_kind, float* mlp, float* mlp_g, float* mlp_m, float* mlp_v, const float* state_f, const int32_t* state_i,
                    float beta1, float beta2, float eps, int enc_pairs, uint16_t* wpack, void* stream) {
    if (n <= 0 || n % 4 != 0) return -1;
    if ((copy_kind != 0) != (table_16 != nullptr) || copy_kind < 0 || copy_kind > 2) return -1;
    const long n4 = (long)(n / 4);
    long blocks = (n4 + 1023) / 1024;
    if (blocks > 256L * 4) blocks = 256L * 4;                 // 4 x 1024-thread blocks per CU, grid-stride beyond
    const dim3 grid((unsigned)blocks + 1), block(1024);
    if (grad_is_f16) {
        if (copy_kind == 2) NGP_ADAM_ALL(2, true); else if (copy_kind == 1) NGP_ADAM_ALL(1, true); else NGP_ADAM_ALL(0, true);
    } else {
        if (copy_kind == 2) NGP_ADAM_ALL(2, false); else if (copy_kind == 1) NGP_ADAM_ALL(1, false); else NGP_ADAM_ALL(0, false);
    }
    NGP_LAUNCH_CHECK();
    return 0;
}
#undef NGP_ADAM_ALL

int ngp_adam_all(float* table, float* table_g, float* table_m, float* table_v, long long n, uint16_t* table_bf16, float* mlp,
                 float* mlp_g, float* mlp_m, float* mlp_v, const float* state_f, const int32_t* state_i, float beta1, float beta2,
                 float eps, int enc_pairs, uint16_t* wpack, void* stream) {
    return ngp_adam_all_ex(table, table_g, 0, table_m, table_v, n, table_bf16, table_bf16 ? 1 : 0, mlp, mlp_g, mlp_m, mlp_v, state_f,
                           state_i, beta1, beta2, eps, enc_pairs, wpack, stream);
}

// Persistent forward grid = what is RESIDENT at once: 150 VGPRs -> 3 waves per SIMD -> three 256-thread blocks per CU = 768 blocks
// (the density-only variant fits 5 per CU; 768 blocks still give it 12 waves per CU).  The launch is sized for the buffer
// capacity (the live count is on the device), and with the former 2048 blocks a training step's ~370 k samples were 1.4 trips per
// wave in 2.7 rounds of block residency, each block reloading the 20 KB weight image for 4-8 tiles.
static inline int mlp_grid(int S) {
    static const int cap = [] { const char* e = ngp_experiment("mlp_fwd_blocks"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
    const int iters = (S + 31) / 32;
    int blocks = (iters + 3) / 4;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

int ngp_mlp_fwd_ex(const float* enc, const float* dirs, const uint16_t* wpack, int n_max, const int32_t* n_dev, int enc_pairs,
                   float* sigmas, uint16_t* rgbs, void* stream) {
    if (n_max <= 0) return 0;
#define NGP_FWD(C, P, D, R)                                                                                          \
    hipLaunchKernelGGL((mlp_fwd_kernel<C, P>), dim3(mlp_grid(n_max)), dim3(256), 0, (hipStream_t)stream, enc, (const float*)(D), \
                       (const half_t*)wpack, n_max, n_dev, sigmas, (half_t*)(R))
    if (dirs && rgbs) { if (enc_pairs) NGP_FWD(true, true, dirs, rgbs); else NGP_FWD(true, false, dirs, rgbs); }
    else { if (enc_pairs) NGP_FWD(false, true, nullptr, nullptr); else NGP_FWD(false, false, nullptr, nullptr); }
#undef NGP_FWD
    NGP_LAUNCH_CHECK();
    return 0;
}

// The forward over a LIST of samples (round 5: the chunked forward): position j < *n_list shades sample list[j].  Entries must be
// valid rows (< n_max) -- including any beyond *n_list that a lane past the end may touch as entry 0.
int ngp_mlp_fwd_list(const float* enc, const float* dirs, const uint16_t* wpack, int n_max, const int32_t* n_list, const int32_t* list,
                     int enc_pairs, float* sigmas, uint16_t* rgbs, void* stream) {
    if (n_max <= 0) return 0;
    if (!n_list || !list || !dirs || !rgbs) return -1;
    if (enc_pairs) hipLaunchKernelGGL((mlp_fwd_kernel<true, true, true>), dim3(mlp_grid(n_max)), dim3(256), 0, (hipStream_t)stream, enc, dirs,
                                      (const half_t*)wpack, n_max, n_list, sigmas, (half_t*)rgbs, list);
    else hipLaunchKernelGGL((mlp_fwd_kernel<true, false, true>), dim3(mlp_grid(n_max)), dim3(256), 0, (hipStream_t)stream, enc, dirs,
                            (const half_t*)wpack, n_max, n_list, sigmas, (half_t*)rgbs, list);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_mlp_fwd(const float* enc, const float* dirs, const uint16_t* wpack, int n, float* sigmas, uint16_t* rgbs, void* stream) {
    return ngp_mlp_fwd_ex(enc, dirs, wpack, n, nullptr, 0, sigmas, rgbs, stream);
}

// launches the backward; parts != NULL: per-block slabs instead of atomics.  Returns the number of blocks (= slabs), < 0 on error
static int mlp_bwd_launch(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs,
                          int n_max, const int32_t* n_dev, const int32_t* live_idx, int enc_pairs, float* d_enc, float* dW,
                          float* parts, int32_t* found_inf, void* stream) {
    // the LDS-image form (12-wave blocks).  -DNGP_MLP_BWD_REG builds also carry round 4's register-resident form (a recorded
    // negative: profiles/microbench/mlp_bwd_reg_kernel.inc), selected per call with NGP_EXPERIMENT mlp_bwd=reg
    int blocks;
#ifdef NGP_MLP_BWD_REG
    const char* form = ngp_experiment("mlp_bwd");
    if (form && form[0] == 'r') {
        if ((long long)n_max * 128 > 0xffffffffLL) return -1;   // (d_enc is addressed through a 32-bit buffer descriptor)
        blocks = ((n_max + 31) / 32 + RW - 1) / RW;
        if (blocks > 256) blocks = 256;                       // one 4-wave block per CU: 1 wave per SIMD (160 accumulator registers)
        if (blocks < 1) blocks = 1;
#define NGP_BWD_REG(P, L)                                                                                                        \
    hipLaunchKernelGGL((mlp_bwd_reg_kernel<P, L>), dim3(blocks), dim3(64 * RW), 0, (hipStream_t)stream, enc, dirs, (const half_t*)wpack, \
                       dsigmas, (const half_t*)drgbs, n_max, n_dev, live_idx, d_enc, dW, parts, found_inf)
        if (enc_pairs) { if (live_idx) NGP_BWD_REG(true, true); else NGP_BWD_REG(true, false); }
        else { if (live_idx) NGP_BWD_REG(false, true); else NGP_BWD_REG(false, false); }
#undef NGP_BWD_REG
        NGP_LAUNCH_CHECK();
        return blocks;
    }
#endif
    blocks = ((n_max + 31) / 32 + BG - 1) / BG;
    if (blocks > 256) blocks = 256;                           // one 12-wave block per CU: 3 waves per SIMD
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(blocks), dim3(64 * BW), 0, (hipStream_t)stream, enc, dirs, (const half_t*)wpack, dsigmas,
                       (const half_t*)drgbs, n_max, n_dev, live_idx, enc_pairs, d_enc, dW, parts, found_inf);
    NGP_LAUNCH_CHECK();
    return blocks;
}

int ngp_mlp_bwd_live(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs,
                     int n_max, const int32_t* n_dev, const int32_t* live_idx, int enc_pairs, float* d_enc, float* dW,
                     int32_t* found_inf, void* stream) {
    if (n_max <= 0) return 0;
    const int rc = mlp_bwd_launch(enc, dirs, wpack, dsigmas, drgbs, n_max, n_dev, live_idx, enc_pairs, d_enc, dW, nullptr, found_inf,
                                  stream);
    return rc < 0 ? rc : 0;
}

int ngp_mlp_dw_parts_max(void) { return 256; }

int ngp_mlp_bwd_live_parts(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs,
                           int n_max, const int32_t* n_dev, const int32_t* live_idx, int enc_pairs, float* d_enc, float* dW_parts,
                           int32_t* found_inf, void* stream) {
    if (n_max <= 0) return 0;
    if (!dW_parts) return -1;
    return mlp_bwd_launch(enc, dirs, wpack, dsigmas, drgbs, n_max, n_dev, live_idx, enc_pairs, d_enc, nullptr, dW_parts, found_inf,
                          stream);
}

__global__ void __launch_bounds__(NGP_MLP_REDUCE_THREADS) mlp_dw_reduce_kernel(const float* __restrict__ parts, int n_parts, float* __restrict__ dW) {
    mlp_dw_reduce_block(parts, n_parts, dW, blockIdx.x);
}

int ngp_mlp_dw_reduce(const float* dW_parts, int n_parts, float* dW, void* stream) {
    if (n_parts <= 0) return 0;
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(NGP_MLP_REDUCE_BLOCKS), dim3(NGP_MLP_REDUCE_THREADS), 0, (hipStream_t)stream, dW_parts, n_parts, dW);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_mlp_bwd_ex(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs,
                   int n_max, const int32_t* n_dev, int enc_pairs, float* d_enc, float* dW, int32_t* found_inf, void* stream) {
    return ngp_mlp_bwd_live(enc, dirs, wpack, dsigmas, drgbs, n_max, n_dev, nullptr, enc_pairs, d_enc, dW, found_inf, stream);
}

int ngp_mlp_bwd(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs, int n,
                float* d_enc, float* dW, void* stream) {
    return ngp_mlp_bwd_ex(enc, dirs, wpack, dsigmas, drgbs, n, nullptr, 0, d_enc, dW, nullptr, stream);
}

}  // extern "C"
