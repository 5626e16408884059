// lds_atomics.hip -- what does an LDS float atomic cost on gfx950?  One 1024-thread block per CU; every wave issues REPS
// ds_add_f32 (or ds_add_u32 / plain ds_write_b32 / ds_add_rtn) instructions with a chosen lane pattern:
//   full-distinct : 64 lanes, 64 different banks           sparse8 : 8 active lanes (exec mask)        same : all lanes one address
//   random        : 64 lanes, random addresses in 128 KB   pair    : lanes (2k, 2k+1) adjacent floats
// Reports cycles per wave-instruction per CU (s_memtime) and G lane-ops/s chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(1024) k(const unsigned* __restrict__ idx, int active, int reps, float* out, unsigned long long* cyc) {
    __shared__ float lds[32768];
    for (int j = threadIdx.x; j < 32768; j += 1024) lds[j] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned a = idx[threadIdx.x];
    const bool on = lane < active;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        const unsigned ad = (a + 97u * r) & 32767u;
        if (on) {
            if (OP == 0) atomicAdd(&lds[ad], 1.0f);
            else if (OP == 1) atomicAdd((unsigned*)&lds[ad], 1u);
            else if (OP == 2) lds[ad] = (float)r;
            else if (OP == 3) acc += __float_as_uint(atomicAdd(&lds[ad], 1.0f));          // returning form
            else if (OP == 4) atomicAdd((unsigned long long*)&lds[ad & 32766u], 0x100000001ull);   // ds_add_u64 (8-byte aligned)
            else if (OP == 6) atomicAdd((double*)&lds[ad & 32766u], 1.0);                          // ds_add_f64
            else if (OP == 5) { unsigned o = atomicAdd((unsigned*)&lds[ad & 32766u], 0x80000001u); atomicAdd((unsigned*)&lds[(ad & 32766u) + 1], 1u + (o + 0x80000001u < o)); }
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int j = threadIdx.x; j < 32768; j += 1024) s += lds[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s + (float)acc;
}

int main() {
    const int blocks = 256, reps = 2000;
    unsigned* idx; float* out; unsigned long long* cyc;
    hipMalloc(&idx, 1024 * 4); hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<unsigned> h(1024);
    std::vector<unsigned long long> hc(blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* opn[7] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_add_rtn_f32", "ds_add_u64", "u32 lo(rtn)+hi carry", "ds_add_f64"};
    struct Pat { const char* name; int mode; int active; };
    Pat pats[] = {{"full-distinct", 0, 64}, {"random", 1, 64}, {"pair-adjacent", 2, 64}, {"sparse8 (8 lanes, random)", 1, 8},
                  {"sparse1 (1 lane)", 1, 1}, {"same-address x64", 3, 64}, {"4 addresses x16", 4, 64}, {"2-lane runs", 5, 64}};
    for (int op = 6; op < 7; ++op)
        for (auto& p : pats) {
            srand(7);
            for (int t = 0; t < 1024; ++t) {
                const int lane = t & 63, wave = t >> 6;
                unsigned a;
                switch (p.mode) {
                    case 0: a = lane + 64 * wave; break;
                    case 1: a = rand() & 32767; break;
                    case 2: a = ((rand() & 16383) << 1); if (lane & 1) a = h[t - 1] + 1; break;
                    case 3: a = 5 * wave; break;
                    case 4: a = 1000 * wave + 37 * (lane >> 4); break;
                    default: a = (lane & 1) ? h[t - 1] : (rand() & 32767); break;
                }
                h[t] = a;
            }
            hipMemcpy(idx, h.data(), 4096, hipMemcpyHostToDevice);
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (op == 0) k<0><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else if (op == 1) k<1><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else if (op == 2) k<2><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else if (op == 3) k<3><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else if (op == 4) k<4><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else if (op == 5) k<5><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                else k<6><<<blocks, 1024>>>(idx, p.active, reps, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            const double instr_per_cu = 16.0 * reps;
            printf("%-15s %-28s %8.3f ms  %7.1f ns per wave-instr per CU  (%6.1f G lane-ops/s chip)  [counter %llu ticks]\n", opn[op], p.name, best,
                   best * 1e6 / instr_per_cu, (double)blocks * 16 * reps * p.active / best / 1e6, hc[0]);
        }
    return 0;
}
