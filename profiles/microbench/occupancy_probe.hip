// How many workgroups of a given shape (threads, LDS bytes, VGPRs) does a gfx950 CU really hold at once?  Every block stamps
// (start, end) on the 100 MHz wall clock and its (XCC, SE/SH/CU) id, spins ~30 us; the host counts the maximum overlap per CU.
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

template <int VGPRS>
__device__ __forceinline__ void claim_vgprs() {
    if (VGPRS >= 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if (VGPRS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    else if (VGPRS >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
}

template <int THREADS, int VGPRS>
__global__ void __launch_bounds__(THREADS) probe(unsigned long long* out, int spin_ticks) {
    extern __shared__ unsigned char lds[];
    claim_vgprs<VGPRS>();
    lds[threadIdx.x] = 1;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) { }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[3 * blockIdx.x] = t0; out[3 * blockIdx.x + 1] = t1;
        out[3 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xffffff00u);
    }
}

template <int THREADS, int VGPRS>
static void run(int lds_bytes, int blocks) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * blocks);
    hipFuncSetAttribute((const void*)probe<THREADS, VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<THREADS, VGPRS>, THREADS, lds_bytes);
    hipLaunchKernelGGL((probe<THREADS, VGPRS>), dim3(blocks), dim3(THREADS), lds_bytes, 0, d, 3000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(3 * blocks);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * blocks, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < blocks; ++b) {
        ev[h[3 * b + 2]].push_back({h[3 * b], +1});
        ev[h[3 * b + 2]].push_back({h[3 * b + 1], -1});
        tmin = std::min(tmin, h[3 * b]); tmax = std::max(tmax, h[3 * b + 1]);
    }
    int worst = 0, best = 1 << 30;
    for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, mx = 0;
        for (auto& e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
        worst = std::max(worst, mx); best = std::min(best, mx);
    }
    printf("threads %4d  lds %6d B  vgprs >= %3d : API says %d blocks/CU; measured max concurrent per CU %d..%d over %zu CUs; %d blocks took %.1f us\n",
           THREADS, lds_bytes, VGPRS, occ, best, worst, ev.size(), blocks, (tmax - tmin) / 100.0);
    hipFree(d);
}

int main() {
    run<384, 168>(76800, 512);
    run<384, 128>(76800, 512);
    run<384, 64>(76800, 512);
    run<384, 168>(65536, 512);
    run<384, 168>(32768, 512);
    run<256, 168>(76800, 512);
    run<512, 128>(76800, 512);
    run<768, 168>(125000, 512);
    return 0;
}
