// atomics2.hip -- which lane patterns does the gfx950 memory pipeline merge for float atomic adds?
// group g: lanes [k*g, k*g+g) of a wave hit g consecutive floats starting at a random (g*4-byte aligned or
// misaligned) table offset.  Also: split (f0 and f1 in two separate instructions), and duplicate addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void scatter(const unsigned* __restrict__ idx, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        unsafeAtomicAdd(dst + idx[i], 1.0f);
}
// two instructions per lane: entry e -> (2e, 2e+1)
__global__ void scatter_split(const unsigned* __restrict__ idx, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned e = idx[i];
        unsafeAtomicAdd(dst + e, 1.0f);
        unsafeAtomicAdd(dst + e + 1, 1.0f);
    }
}
__global__ void scatter_lds_flush(float* __restrict__ dst, long table, int reps) {   // fully coalesced atomics: 64 consecutive floats
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < table; i += (long)gridDim.x * blockDim.x)
            unsafeAtomicAdd(dst + i, 1.0f);
}

int main() {
    const long n = 48L << 20;
    const long table = 11420064;
    unsigned* idx; float* dst;
    hipMalloc(&idx, n * 4); hipMalloc(&dst, table * 4);
    std::vector<unsigned> h(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, long count, int mode) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(dst, 0, table * 4);
            hipEventRecord(e0);
            if (mode == 0) scatter<<<4096, 256>>>(idx, dst, count);
            else if (mode == 1) scatter_split<<<4096, 256>>>(idx, dst, count);
            else scatter_lds_flush<<<4096, 256>>>(dst, table, 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        long atoms = mode == 1 ? 2 * count : (mode == 2 ? 4 * table : count);
        printf("%-44s %.3f ms  %.1f Gatom/s\n", name, best, atoms / best / 1e6);
    };
    char name[128];
    for (int g = 1; g <= 64; g *= 2) {
        for (int mis = 0; mis < 2; ++mis) {
            if (g == 1 && mis) continue;
            srand(1);
            for (long i = 0; i < n; i += g) {
                unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
                unsigned base = (r % (unsigned)((table - 2 * g) / g)) * g + (mis ? g / 2 : 0);
                for (int k = 0; k < g && i + k < n; ++k) h[i + k] = base + k;
            }
            hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
            snprintf(name, sizeof name, "group of %2d adjacent lanes, %s", g, mis ? "misaligned by g/2" : "aligned");
            run(name, n, 0);
        }
    }
    // duplicates inside an instruction: groups of g lanes hit the SAME address
    for (int g = 2; g <= 16; g *= 2) {
        srand(1);
        for (long i = 0; i < n; i += g) {
            unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
            for (int k = 0; k < g && i + k < n; ++k) h[i + k] = r % table;
        }
        hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
        snprintf(name, sizeof name, "group of %2d lanes, same address", g);
        run(name, n, 0);
    }
    srand(1);
    for (long i = 0; i < n / 2; ++i) { unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand(); h[i] = (r % (table / 2 - 1)) * 2; }
    hipMemcpy(idx, h.data(), n * 2, hipMemcpyHostToDevice);
    run("split: f0,f1 as two instructions per lane", n / 2, 1);
    run("dense sweep: 64 consecutive floats / wave", 0, 2);
    return 0;
}
