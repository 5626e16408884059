// exchange.hip -- the gradient exchange of SURVEY section 8(e) as DIRECT peer-memory writes over xGMI (round 6 prototype).
//
// The reference has no multi-GPU path; north_star asks for "rays shard across the GPUs with an all-reduce of the hash-table and MLP
// gradients per step".  The default form (ngp_hip/trainer.py) is RCCL: reduce-scatter -> Adam on the own 1/N of the table -> all-gather.
// On xGMI that is a ring (7 point-to-point links per GPU, no switch): 2 (N - 1) / N x 45.7 MB per rank in 2 (N - 1) latency-bound hops,
// ~0.5 ms at 8 GPUs.  The direct form uses every link at once, ONE hop per phase:
//   push  : rank r writes slice p of its gradient straight into row r of rank p's INBOX (peer memory mapped through hipIpc), for all p,
//           then raises flag[r] in rank p's flag word block (system-scope release behind the data);
//   wait  : rank p waits (ONE small workgroup polling with system-scope acquire, bounded) for all N flags of this step, then a second
//           launch reduces its N inbox rows to the
//           average gradient of ITS shard -- the reduce-scatter, with 1 / N of the table on each of the N - 1 links: ~40 MB out over
//           7 links x ~150 GB/s = ~40 us at 8 GPUs (SURVEY section 5's 75 us for both phases);
//   after Adam on the shard, the updated parameters travel back the same way (push of the own shard into every peer's table copy +
//   flags + wait): the all-gather.
// One-GPU functional evidence only (tests/test_gpu_p2p.py: two / three processes on ONE device, every "peer" pointer an IPC mapping of
// another process's allocation); no multi-GPU node has been available in six rounds, so nothing here has met xGMI -- in particular the
// system-scope loads of the inbox below are what CORRECTNESS on coarse-grained peer-written memory needs, not a measured choice.
#include "ngp_device.h"

namespace ngp {

constexpr int XCH_MAX_PEERS = 16;

struct PeerPtrs {
    void* p[XCH_MAX_PEERS];          // peer k's destination base (inbox / table copy), mapped into this process
    int32_t* flag[XCH_MAX_PEERS];    // peer k's flag block (one word per source rank)
};

// dst_k[dst_off + i] = src[k * n_per_peer + i]  (slice k goes to peer k), 16 bytes per lane; the last block out raises the flags.
// bcast != 0: the SAME n_per_peer elements (src[0 .. n_per_peer)) go to every peer (the all-gather phase).
__global__ void __launch_bounds__(256) p2p_push_kernel(const uint4* __restrict__ src, long n16_per_peer, int world, PeerPtrs peers,
                                                       long dst_off16, int bcast, int rank, int step, uint32_t* __restrict__ done) {
    const long total = n16_per_peer * world;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / n16_per_peer);
        const long j = i - (long)k * n16_per_peer;
        const uint4 v = src[bcast ? j : i];
        reinterpret_cast<uint4*>(peers.p[k])[dst_off16 + j] = v;
    }
    __threadfence_system();                                  // this block's data is out before its "done" increment
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(done, 1u) == gridDim.x - 1u) {          // last block: every block's writes precede this point (fence above)
            *done = 0u;
            __threadfence_system();
            for (int k = 0; k < world; ++k)
                __hip_atomic_store(peers.flag[k] + rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Bounded wait for flags[0 .. world) >= step: ONE workgroup, one polling lane per peer (s_sleep between polls).  A single small workgroup on
// purpose: while it waits, the peers' kernels must be able to run -- on the multi-GPU node they are on other devices anyway, but the
// one-device functional test time-shares ONE device between the ranks, and a full grid of spinning workgroups would keep the very kernels
// it waits for off the CUs until the spin budget is gone.  A wait that runs out of polls raises err[0].
__global__ void __launch_bounds__(64) p2p_wait_kernel(const int32_t* __restrict__ flags, int world, int step, long max_spins,
                                                      int32_t* __restrict__ err) {
    const int k = threadIdx.x;
    bool good = k >= world;
    for (long s = 0; s < max_spins; ++s) {
        if (!good) good = __hip_atomic_load(flags + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= step;
        if (__all(good)) break;
        __builtin_amdgcn_s_sleep(64);
    }
    if (!__all(good) && k == 0) atomicExch(err, 1);
    __threadfence_system();
}

// out[i] = scale * sum_k inbox[k * n + i], in rank order (the same sum on every rank that reduces the same rows); skipped when the wait
// in front of it gave up.
__global__ void __launch_bounds__(256) p2p_reduce_kernel(const float* __restrict__ inbox, int world, long n, float scale,
                                                         float* __restrict__ out, const int32_t* __restrict__ err) {
    if (*err) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < world; ++k)       // (system-scope loads: the rows were written by OTHER devices into coarse-grained memory)
            acc += __hip_atomic_load(inbox + (long)k * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[i] = acc * scale;
    }
}

}  // namespace ngp

using namespace ngp;

extern "C" {

int ngp_p2p_max_peers(void) { return XCH_MAX_PEERS; }

// peer_dst / peer_flags: `world` device pointers each (host arrays).  n_per_peer elements of elem_bytes (4 or 2) per peer, a multiple
// of 16 bytes; dst_offset in ELEMENTS inside each peer's destination.  done: one uint32 of this rank's own memory, 0 between launches.
int ngp_p2p_push(const void* src, long long n_per_peer, int elem_bytes, int world, void* const* peer_dst, int32_t* const* peer_flags,
                 long long dst_offset, int broadcast, int rank, int step, uint32_t* done, void* stream) {
    if (!src || !peer_dst || !peer_flags || !done || world < 1 || world > XCH_MAX_PEERS || rank < 0 || rank >= world) return -1;
    if (elem_bytes != 4 && elem_bytes != 2) return -1;
    if ((n_per_peer * elem_bytes) % 16 != 0 || (dst_offset * elem_bytes) % 16 != 0 || n_per_peer <= 0) return -1;
    PeerPtrs pp = {};
    for (int k = 0; k < world; ++k) { pp.p[k] = peer_dst[k]; pp.flag[k] = peer_flags[k]; if (!pp.p[k] || !pp.flag[k]) return -1; }
    const long n16 = (long)(n_per_peer * elem_bytes / 16);
    long blocks = (n16 * world + 255) / 256;
    // one workgroup per CU at most: every block ends in a system-scope release, and this round measured what many of those cost on this
    // part (csrc/optim.hip, check_finite_prologue_kernel: 4096 agent-scope fences behind a kernel that left the L2s dirty = 260 us);
    // 256 blocks of 16-byte stores are more than seven xGMI links take
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(p2p_push_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, n16, world, pp,
                       (long)(dst_offset * elem_bytes / 16), broadcast, rank, step, done);
    NGP_LAUNCH_CHECK();
    return 0;
}

// flags: this rank's own flag block (`world` words).  reduce != 0: out[0 .. n) = scale * sum over the `world` rows of inbox ([world, n] f32).
int ngp_p2p_wait(const int32_t* flags, int world, int step, long long max_spins, const float* inbox, long long n, float scale, int reduce,
                 float* out, int32_t* err, void* stream) {
    if (!flags || !err || world < 1 || world > XCH_MAX_PEERS || max_spins <= 0) return -1;
    if (reduce && (!inbox || !out || n <= 0)) return -1;
    hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flags, world, step, (long)max_spins, err);
    if (reduce) {
        long blocks = (n + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(p2p_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, inbox, world, (long)n, scale, out, err);
    }
    NGP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
