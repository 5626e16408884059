// What does one scattered gather cost the vector L1 (TCP) on gfx950?  The forward hash gather issues 8 independent 8-byte loads per
// (sample, level), every lane to a different cache line, and r04_pmc.json shows ~1 TCP access per clock per CU at its speed.
// This probe times the same pattern in isolation on a table that every XCD's L2 holds (2 MB), so the TCP -> L2 path is what is
// measured, and varies the access: width (4 / 8 / 16 bytes per lane), 16-byte loads at 8-byte alignment, and instructions where
// half of the lanes are switched off -- by a buffer range check (offset pushed out of range) or by a divergent branch.
//   hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: b32   1: b64   2: b128 16-aligned   3: b128 8-aligned   4: b64, odd lanes dropped by the range check
//      5: b64, odd lanes skipped by a branch   6: b128 8-aligned on even lanes (branch) + 2 x b64 on odd lanes (branch)
template <int MODE>
__global__ void __launch_bounds__(256) probe(const unsigned* __restrict__ table, unsigned bytes, int iters, unsigned* __restrict__ sink) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, (int)bytes, 0x00020000);
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const bool odd = threadIdx.x & 1;
    for (int it = 0; it < iters; ++it) {
        unsigned off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) off[k] = lcg(s) % (bytes - 64u);
        if constexpr (MODE == 0) {
            unsigned v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[k] & ~3u, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k];
        } else if constexpr (MODE == 1 || MODE == 4) {
            uintx2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (MODE == 4 && odd) ? 0xffffffffu : (off[k] & ~7u), 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y;
        } else if constexpr (MODE == 2 || MODE == 3) {
            uintx4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, MODE == 2 ? (off[k] & ~15u) : ((off[k] & ~15u) | 8u), 0, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        } else if constexpr (MODE == 5) {
            if (!odd) {
                uintx2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[k] & ~7u, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y;
            }
        } else {
            if (!odd) {
                uintx4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (off[k] & ~15u) | 8u, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
            } else {
                uintx2 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off[k] & ~7u, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* what, const unsigned* table, unsigned bytes, unsigned* sink, double lanes_per_thread_iter) {
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, table, bytes, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double accesses = (double)blocks * 256 * iters * lanes_per_thread_iter;      // lane-accesses that reach the cache
    printf("%-78s %8.1f us   %6.3f lane-accesses / ns / CU   (%5.2f per clock at 2.4 GHz)\n", what, best * 1e3,
           accesses / (best * 1e6) / 256.0, accesses / (best * 1e6) / 256.0 / 2.4);
}

int main() {
    const unsigned sizes[3] = {2u << 20, 16u << 10, 256u << 10};
    const char* names[3] = {"2 MB table: misses the 32 KB vector L1, hits every XCD's L2", "16 KB table: vector-L1 hits", "256 KB table: L1 misses, L2 hits, denser"};
    for (int t = 0; t < 3; ++t) {
        const unsigned bytes = sizes[t];
        unsigned *table, *sink;
        hipMalloc(&table, bytes); hipMalloc(&sink, 4);
        std::vector<unsigned> h(bytes / 4);
        for (auto& x : h) x = (unsigned)rand();
        hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice);
        printf("---- %s\n", names[t]);
        run<0>("b32, all lanes", table, bytes, sink, 8);
        run<1>("b64, all lanes", table, bytes, sink, 8);
        run<2>("b128 at 16-byte alignment, all lanes", table, bytes, sink, 8);
        run<3>("b128 at 8-byte alignment (may straddle a line), all lanes", table, bytes, sink, 8);
        run<4>("b64, odd lanes dropped by the buffer range check (live lanes counted)", table, bytes, sink, 4);
        run<5>("b64, odd lanes skipped by a divergent branch (live lanes counted)", table, bytes, sink, 4);
        run<6>("even lanes: 4 x b128 at 8-byte alignment; odd lanes: 8 x b64 (two branches)", table, bytes, sink, 6);
        hipFree(table); hipFree(sink);
    }
    return 0;
}
