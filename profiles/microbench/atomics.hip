// atomics.hip -- microbenchmark: float atomic-add throughput on MI355X by scope and address pattern.
// Decides the design of the hash-grid backward (45.7 MB table, 256 atomics / sample).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int SCOPE>
__global__ void scatter(const unsigned* __restrict__ idx, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned j = idx[i];
        if (SCOPE == 0) unsafeAtomicAdd(dst + j, 1.0f);                                                   // agent scope
        else if (SCOPE == 1) __hip_atomic_fetch_add(dst + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(dst + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else dst[j] += 1.0f;                                                                               // racy RMW (upper bound)
    }
}

int main() {
    const long n = 48L << 20;          // ~ one hash backward at 190k samples
    const long table = 11420064;
    unsigned *idx; float* dst;
    hipMalloc(&idx, n * 4); hipMalloc(&dst, table * 4);
    std::vector<unsigned> h(n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* pat[] = {"uniform-45MB", "uniform-4MB(one level)", "hot-4096", "runs-of-32-same-addr", "pairs(adjacent f0,f1)"};
    for (int p = 0; p < 5; ++p) {
        srand(1);
        for (long i = 0; i < n; ++i) {
            unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
            if (p == 0) h[i] = r % table;
            else if (p == 1) h[i] = r % (1 << 20);
            else if (p == 2) h[i] = r % 4096;
            else if (p == 3) { if (i % 32 == 0) h[i] = r % table; else h[i] = h[i - 1]; }
            else { if (i % 2 == 0) h[i] = (r % (table / 2)) * 2; else h[i] = h[i - 1] + 1; }
        }
        hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int s = 0; s < 4; ++s) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(dst, 0, table * 4);
                hipEventRecord(e0);
                if (s == 0) scatter<0><<<4096, 256>>>(idx, dst, n);
                if (s == 1) scatter<1><<<4096, 256>>>(idx, dst, n);
                if (s == 2) scatter<2><<<4096, 256>>>(idx, dst, n);
                if (s == 3) scatter<3><<<4096, 256>>>(idx, dst, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            // verify the sum for atomic variants
            std::vector<float> out(table); hipMemcpy(out.data(), dst, table * 4, hipMemcpyDeviceToHost);
            double sum = 0; for (long i = 0; i < table; ++i) sum += out[i];
            printf("%-26s scope=%d  %.3f ms  %.1f Gatom/s  sum/n=%.6f\n", pat[p], s, best, n / best / 1e6, sum / n);
        }
    }
    return 0;
}
