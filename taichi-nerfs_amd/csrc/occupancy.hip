// occupancy.hip -- the occupancy-grid update of Instant-NGP as device-resident kernels for gfx950.
//
// Replaces the torch-op sequence of reference modules/networks.py:181-209 (sample_uniform_and_occupied_cells) and
// :255-290 (update_density_grid): ~45 small launches, a torch.nonzero (host sync, dynamic shape) and a .item() (host sync)
// every 16 training steps, during which the GPU idles.  Here the update is a fixed sequence of launches with no
// read-back: the list of occupied cells is compacted on the device, sampled through its device-side count, the mean
// density is reduced on the device and consumed by the packbits kernel from device memory.
// The density evaluation in the middle reuses the hash-grid and fused-MLP kernels.
#include "ngp_device.h"

namespace ngp {

// cells of one cascade with density > threshold -> list[0 .. count) (order unspecified; they are sampled uniformly)
// Deterministic three-launch compaction (count per wave chunk -> one-block exclusive scan -> write): the list comes out in
// cell order on every run and every rank (an atomic slice reservation would order the chunks by arrival), and nothing
// serialises on a same-address atomic (~12 ns each: one per 64 cells once cost 390 us for 128^3 cells).
constexpr int OCC_WAVES = 1024;            // 256 blocks x 4 waves; also the scan block's width
__device__ __forceinline__ void occ_chunk(int n_cells, int& lo, int& hi) {
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int chunk = ((n_cells + OCC_WAVES - 1) / OCC_WAVES + 63) & ~63;
    lo = min(wave * chunk, n_cells); hi = min(lo + chunk, n_cells);
}
__global__ void __launch_bounds__(256) occ_count_kernel(const float* __restrict__ grid, float thr, int n_cells,
                                                        int32_t* __restrict__ wave_counts) {
    int lo, hi;
    occ_chunk(n_cells, lo, hi);
    const int lane = threadIdx.x & 63;
    int mine = 0;
    for (int i = lo + lane; i < hi; i += 64) mine += grid[i] > thr ? 1 : 0;                  // networks.py:198-199
    const int total = wave_sum_i(mine);
    if (lane == 0) wave_counts[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = total;
}
__global__ void __launch_bounds__(OCC_WAVES) occ_scan_kernel(int32_t* __restrict__ wave_counts, int32_t* __restrict__ count) {
    __shared__ int part[OCC_WAVES / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int c = wave_counts[t];
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
    if (lane == 63) part[w] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += part[k];
    wave_counts[t] = base + incl - c;                                                        // exclusive prefix = the chunk's slice start
    if (t == OCC_WAVES - 1) count[0] = base + incl;
}
__global__ void __launch_bounds__(256) occ_write_kernel(const float* __restrict__ grid, float thr, int n_cells,
                                                        const int32_t* __restrict__ wave_base, int32_t* __restrict__ list) {
    int lo, hi;
    occ_chunk(n_cells, lo, hi);
    const int lane = threadIdx.x & 63;
    int base = wave_base[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool occ = i < hi && grid[i] > thr;
        const unsigned long long m = __ballot(occ);
        if (occ) list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        base += __popcll(m);
    }
}

// M uniform cells + M cells drawn (with replacement) from the occupied list, their Morton indices and a jittered world
// position inside each cell (networks.py:193-207, :270-275).
__global__ void __launch_bounds__(256) occ_sample_kernel(const float* __restrict__ u_cell /*[M]*/, const float* __restrict__ u_pick /*[M]*/,
                                                         const float* __restrict__ u_jit /*[2M,3]*/, const int32_t* __restrict__ list,
                                                         const int32_t* __restrict__ count, int M, int G, float s, float half_grid,
                                                         int32_t* __restrict__ indices /*[2M]*/, float* __restrict__ xyzs /*[2M,3]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * M) return;
    int cx, cy, cz, idx;
    if (i < M) {
        // torch.randint(grid_size, (M,3)) + morton3D (:193-196) == one uniform Morton code: the code is a bijection of the
        // cell coordinates, so drawing it directly is the same distribution.  The caller hands the uniforms in ASCENDING
        // order (order statistics of M iid uniforms), which makes consecutive encoder queries spatially coherent.
        const int G3 = G * G * G;
        idx = min((int)((double)u_cell[i] * (double)G3), G3 - 1);
        cx = morton3d_invert1((uint32_t)idx); cy = morton3d_invert1((uint32_t)idx >> 1); cz = morton3d_invert1((uint32_t)idx >> 2);
    } else {
        const int n_occ = *count;
        if (n_occ > 0) {                                                           // :200-203
            const int k = min((int)(u_pick[i - M] * (float)n_occ), n_occ - 1);
            idx = list[k];
        } else {
            idx = 0;                                                               // torch: empty index list -> no cells; cell 0 is
        }                                                                          // a harmless stand-in (only ever max-merged)
        cx = morton3d_invert1((uint32_t)idx); cy = morton3d_invert1((uint32_t)idx >> 1); cz = morton3d_invert1((uint32_t)idx >> 2);
    }
    indices[i] = idx;
    const int c[3] = {cx, cy, cz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float base = ((float)c[k] / (float)(G - 1) * 2.0f - 1.0f) * (s - half_grid);      // :272-273
        xyzs[3 * (size_t)i + k] = base + (u_jit[3 * (size_t)i + k] * 2.0f - 1.0f) * half_grid;  // :275
    }
}

// warm-up variant: every cell, enumerated in Morton order (index i IS the Morton code)
__global__ void __launch_bounds__(256) occ_all_cells_kernel(const float* __restrict__ u_jit /*[n,3]*/, int n_cells, int G, float s,
                                                            float half_grid, float* __restrict__ xyzs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const int c[3] = {morton3d_invert1((uint32_t)i), morton3d_invert1((uint32_t)i >> 1), morton3d_invert1((uint32_t)i >> 2)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float base = ((float)c[k] / (float)(G - 1) * 2.0f - 1.0f) * (s - half_grid);
        xyzs[3 * (size_t)i + k] = base + (u_jit[3 * (size_t)i + k] * 2.0f - 1.0f) * half_grid;
    }
}

// density_grid_tmp[indices] = sigmas (duplicates: any winner, like torch's index_put_ with repeated indices, :276)
__global__ void __launch_bounds__(256) occ_scatter_kernel(const int32_t* __restrict__ indices, const float* __restrict__ sigmas, int n,
                                                          float* __restrict__ tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tmp[indices ? indices[i] : i] = sigmas[i];
}

// deterministic form of the scatter: a cell drawn twice (the uniform and the occupied half overlap, and both halves draw with
// replacement) is written by two threads with different jittered densities, and the reference's fresh[c, indices] = density keeps
// whichever write lands last.  Here the LARGEST wins -- one of the values the race could have kept, the same one on every run
// (densities are exp(.) > 0, whose float order is their bit patterns' integer order; tmp starts at 0).
__global__ void __launch_bounds__(256) occ_scatter_max_kernel(const int32_t* __restrict__ indices, const float* __restrict__ sigmas, int n,
                                                              float* __restrict__ tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = sigmas[i];
    if (s > 0.0f) atomicMax(reinterpret_cast<int*>(tmp) + (indices ? indices[i] : i), __float_as_int(s));
}

// grid = where(grid < 0, grid, max(grid*decay, tmp)) (:281-284) and the sum / count of the positive cells (:286).  Round 5: every
// block leaves its partial (sum, count) in stats[2 + 2 b ..] instead of two float atomics on stats[0..1] -- the mean is the
// occupancy threshold, and a sum whose order changes from run to run moves cells that sit on the threshold in and out of the grid.
constexpr int OCC_MERGE_MAX_BLOCKS = 512;
__host__ __device__ __forceinline__ int occ_merge_blocks(long n) {
    long blocks = (n + 255) / 256;
    blocks = (blocks + 3) / 4;                 // float4 per thread
    return (int)(blocks > OCC_MERGE_MAX_BLOCKS ? OCC_MERGE_MAX_BLOCKS : (blocks < 1 ? 1 : blocks));
}
__global__ void __launch_bounds__(256) occ_merge_kernel(float* __restrict__ grid, const float* __restrict__ tmp, float decay, int n,
                                                        float* __restrict__ stats /*[2 + 2 * 512]: [0] sum, [1] count, then per-block partials*/) {
    float sum = 0.f, cnt = 0.f;
    const int n4 = n >> 2;
    float4* g4 = reinterpret_cast<float4*>(grid);
    const float4* t4 = reinterpret_cast<const float4*>(tmp);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 g = g4[i];
        const float4 t = t4[i];
#define NGP_MERGE1(c) { if (!(g.c < 0.f)) g.c = fmaxf(g.c * decay, t.c); if (g.c > 0.f) { sum += g.c; cnt += 1.f; } }
        NGP_MERGE1(x) NGP_MERGE1(y) NGP_MERGE1(z) NGP_MERGE1(w)
#undef NGP_MERGE1
        g4[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                   // tail (n is a multiple of 4 for 128^3 grids)
        const int i = (n4 << 2) + threadIdx.x;
        float g = grid[i];
        if (!(g < 0.f)) g = fmaxf(g * decay, tmp[i]);
        grid[i] = g;
        if (g > 0.f) { sum += g; cnt += 1.f; }
    }
    sum = wave_sum(sum); cnt = wave_sum(cnt);
    __shared__ float ps[4], pc[4];
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = sum; pc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[2 + 2 * blockIdx.x] = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        stats[3 + 2 * blockIdx.x] = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    }
}

// packbits with threshold = min(mean density, density_threshold) (:286-290).  Every block adds the merge kernel's partials up
// itself, in one fixed order (4 KB from L2; no extra launch, no atomics), block 0 also leaves the totals in stats[0..1].
__global__ void __launch_bounds__(256) occ_pack_kernel(const float4* __restrict__ grid, float* __restrict__ stats, int n_partials, float thr_max,
                                                       int n_bytes, uint8_t* __restrict__ out) {
    __shared__ float ps[4], pc[4];
    float s = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < n_partials; b += 256) { s += stats[2 + 2 * b]; c += stats[3 + 2 * b]; }
    s = wave_sum(s); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    const float total = (ps[0] + ps[1]) + (ps[2] + ps[3]), count = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = total; stats[1] = count; }
    const float thr = fminf(total / count, thr_max);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_bytes; n += gridDim.x * blockDim.x) {
        const float4 a = grid[2 * (size_t)n], b = grid[2 * (size_t)n + 1];
        uint32_t bits = 0;
        bits |= (a.x > thr) ? 1u : 0u; bits |= (a.y > thr) ? 2u : 0u; bits |= (a.z > thr) ? 4u : 0u; bits |= (a.w > thr) ? 8u : 0u;
        bits |= (b.x > thr) ? 16u : 0u; bits |= (b.y > thr) ? 32u : 0u; bits |= (b.z > thr) ? 64u : 0u; bits |= (b.w > thr) ? 128u : 0u;
        out[n] = (uint8_t)bits;
    }
}

// ---- ascending uniforms without sorting ----------------------------------------------------------------------------------
// The order statistics of m iid U(0,1) are distributed like the normalised partial sums of m + 1 unit exponentials
// E_k = -log(1 - U_k): out[k] = (E_0 + ... + E_k) / (E_0 + ... + E_m).  Two launches (row scans of 1024, then row offsets +
// normalisation) instead of the ten small torch kernels the same formula took (rand, neg, log1p, neg, 2 x cumsum, copy, 2 x add,
// div: ~105 us per occupancy update).  u: [sets][rows * 1024] uniforms, work: [sets][rows * 1024 + rows], out: [sets][m].
__global__ void __launch_bounds__(1024) sorted_uniform_rows_kernel(const float* __restrict__ u, int rows, float* __restrict__ work) {
    __shared__ float wsum[16];
    const int row = blockIdx.x, set = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t base = ((size_t)set * rows + row) * 1024;
    float* c1 = work + (size_t)set * (rows * 1024 + rows);
    const float e = -log1pf(-u[base + tid]);
    const float inc = wave_scan_add(e, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    float woff = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) woff += (w < wv) ? wsum[w] : 0.0f;
    c1[(size_t)row * 1024 + tid] = woff + inc;
    if (tid == 1023) c1[(size_t)rows * 1024 + row] = woff + inc;          // row total
}
__global__ void __launch_bounds__(1024) sorted_uniform_norm_kernel(const float* __restrict__ work, int rows, int m, float* __restrict__ out) {
    __shared__ double red[2][16];
    __shared__ double s_off, s_den;
    const int row = blockIdx.x, set = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* c1 = work + (size_t)set * (rows * 1024 + rows);
    const float* tot = c1 + (size_t)rows * 1024;
    // offset of this row, and of the row that holds element m (the normaliser).  Summed in f64, where a sum of <= 2^20 f32 row
    // totals of ~1e3 is EXACT, so every block gets off[r + 1] == off[r] + tot[r] whatever the reduction order and the output is
    // monotone across row boundaries (an f32 tree reduction differs in the last ulp of ~5e5 and stepped back by 1e-7 there; a
    // serial f32 sum by one thread is 513 dependent global loads, +45 us per training step)
    const int rm = m >> 10;
    double voff = 0.0, vden = 0.0;
    for (int r = tid; r < rows; r += 1024) {
        const double t = (double)tot[r];
        if (r < row) voff += t;
        if (r < rm) vden += t;
    }
    for (int o = 32; o > 0; o >>= 1) { voff += __shfl_down(voff, o); vden += __shfl_down(vden, o); }
    if (lane == 0) { red[0][wv] = voff; red[1][wv] = vden; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
        s_off = a;
        s_den = b + (double)c1[(size_t)rm * 1024 + (m & 1023)];
    }
    __syncthreads();
    const int k = row * 1024 + tid;
    if (k < m) out[(size_t)set * m + k] = (float)(((double)c1[(size_t)row * 1024 + tid] + s_off) / s_den);
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_occ_compact(const float* density_grid, float threshold, int n_cells, int32_t* list, int32_t* count, int32_t* scratch,
                    void* stream) {
    if (n_cells <= 0) return 0;
    if (!scratch) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(occ_count_kernel, dim3(OCC_WAVES / 4), dim3(256), 0, s, density_grid, threshold, n_cells, scratch);
    hipLaunchKernelGGL(occ_scan_kernel, dim3(1), dim3(OCC_WAVES), 0, s, scratch, count);
    hipLaunchKernelGGL(occ_write_kernel, dim3(OCC_WAVES / 4), dim3(256), 0, s, density_grid, threshold, n_cells, scratch, list);
    NGP_LAUNCH_CHECK();
    return 0;
}

// out[set][k], k < m: ascending uniforms (see sorted_uniform_rows_kernel); u holds sets * rows * 1024 uniforms, rows = (m + 1 +
// 1023) / 1024; work holds sets * (rows * 1024 + rows) floats
int ngp_sorted_uniforms(const float* u, int m, int sets, float* work, float* out, void* stream) {
    if (m <= 0 || sets <= 0) return 0;
    const int rows = (m + 1 + 1023) / 1024;
    hipLaunchKernelGGL(sorted_uniform_rows_kernel, dim3(rows, sets), dim3(1024), 0, (hipStream_t)stream, u, rows, work);
    hipLaunchKernelGGL(sorted_uniform_norm_kernel, dim3(rows, sets), dim3(1024), 0, (hipStream_t)stream, (const float*)work, rows, m, out);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_sample(const float* u_cell, const float* u_pick, const float* u_jit, const int32_t* list, const int32_t* count, int m,
                   int grid_size, float s, float half_grid, int32_t* indices, float* xyzs, void* stream) {
    if (m <= 0) return 0;
    hipLaunchKernelGGL(occ_sample_kernel, dim3((2 * m + 255) / 256), dim3(256), 0, (hipStream_t)stream, u_cell, u_pick, u_jit, list,
                       count, m, grid_size, s, half_grid, indices, xyzs);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_all_cells(const float* u_jit, int n_cells, int grid_size, float s, float half_grid, float* xyzs, void* stream) {
    if (n_cells <= 0) return 0;
    hipLaunchKernelGGL(occ_all_cells_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, (hipStream_t)stream, u_jit, n_cells, grid_size,
                       s, half_grid, xyzs);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_scatter(const int32_t* indices, const float* sigmas, int n, float* tmp, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(occ_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, indices, sigmas, n, tmp);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_scatter_max(const int32_t* indices, const float* sigmas, int n, float* tmp, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(occ_scatter_max_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, indices, sigmas, n, tmp);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_stats_floats(void) { return 2 + 2 * OCC_MERGE_MAX_BLOCKS; }

int ngp_occ_merge(float* density_grid, const float* tmp, float decay, int n, float* stats, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(occ_merge_kernel, dim3(occ_merge_blocks(n)), dim3(256), 0, (hipStream_t)stream, density_grid, tmp, decay, n, stats);
    NGP_LAUNCH_CHECK();
    return 0;
}

int ngp_occ_pack(const float* density_grid, float* stats, float density_threshold, int n_bytes, uint8_t* bitfield, void* stream) {
    if (n_bytes <= 0) return 0;
    int blocks = (n_bytes + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(occ_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)density_grid, stats,
                       occ_merge_blocks(8L * n_bytes), density_threshold, n_bytes, bitfield);
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
