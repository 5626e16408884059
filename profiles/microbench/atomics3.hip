// atomics3.hip -- does XCD-local placement help float atomics / gathers?  Block b only touches the table slice of
// "XCD" b % 8 (observed block->XCD mapping) vs. every block touching the whole 45 MB table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// quads of adjacent lanes hit 4 consecutive floats (the shape the hash backward issues)
template <bool LOCAL, bool ATOMIC>
__global__ void k(const unsigned* __restrict__ rnd, float* __restrict__ dst, long n, unsigned slice_floats) {
    const unsigned xcd = blockIdx.x & 7u;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned r = rnd[i >> 2];                      // same random number for the 4 lanes of a quad
        unsigned off = LOCAL ? (xcd * slice_floats + (r % (slice_floats / 4)) * 4) : ((r % (slice_floats * 2)) * 4);
        off += (unsigned)(i & 3);
        if (ATOMIC) unsafeAtomicAdd(dst + off, 1.0f); else acc += dst[off];
    }
    if (!ATOMIC && acc == 12345.f) dst[0] = acc;
}

int main() {
    const long n = 48L << 20;
    const unsigned slice_floats = 1u << 20;            // 4 MB per "XCD" slice, 32 MB total
    unsigned* rnd; float* dst;
    hipMalloc(&rnd, (n / 4) * 4); hipMalloc(&dst, (size_t)slice_floats * 8 * 4);
    std::vector<unsigned> h(n / 4);
    srand(3);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    hipMemcpy(rnd, h.data(), (n / 4) * 4, hipMemcpyHostToDevice);
    hipMemset(dst, 0, (size_t)slice_floats * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int mode) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<false, true><<<4096, 256>>>(rnd, dst, n, slice_floats);
            if (mode == 1) k<true, true><<<4096, 256>>>(rnd, dst, n, slice_floats);
            if (mode == 2) k<false, false><<<4096, 256>>>(rnd, dst, n, slice_floats);
            if (mode == 3) k<true, false><<<4096, 256>>>(rnd, dst, n, slice_floats);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-52s %.3f ms  %.1f G lane-ops/s  %.1f G quads(lines)/s\n", name, best, n / best / 1e6, n / 4 / best / 1e6);
    };
    run("atomic quads, whole 32 MB table from every block", 0);
    run("atomic quads, 4 MB slice of XCD (blockIdx % 8)", 1);
    run("gather quads, whole 32 MB table from every block", 2);
    run("gather quads, 4 MB slice of XCD (blockIdx % 8)", 3);
    return 0;
}
