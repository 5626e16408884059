/*
 * ngp_hip.h -- C ABI of libngp_hip.so: the MI355X (gfx950) Instant-NGP training hot path.
 *
 * This is the drop-in boundary under the Python surface of the reference's `modules/` package
 * (taichi-dev/taichi-nerfs).  Every entry point replaces ONE Taichi kernel launch of the reference;
 * the reference file:line each one stands in for is cited next to it.  The reference binds its kernels
 * through Taichi's torch zero-copy interop (tensor -> raw device pointer); the binding a maintainer adds
 * instead is the ctypes stub shown in INTEGRATION.md (it is what taichi-nerfs_amd/ngp_hip/lib.py does).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + explicit sizes; no torch types; no allocation; no host sync
 *     (the only exception is ngp_march_train_total, which is an explicit D2H read).
 *   - every function returns 0 on success, or the negated hipError_t of the failing launch.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - all geometric data is f32, `rays_a` is i32, alive/ray indices are i64 -- as in the reference.
 *   - arrays are dense row-major ("contiguous" in torch terms); the reference wrappers call
 *     .contiguous() before every launch (hash_encoder.py:283, volume_train.py:188, rendering.py:34).
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_MAX_LEVELS 16
#define NGP_ABI_VERSION 3

/* Multiresolution level table.  Built ONCE on the host (ngp_hash_levels_init) in the arithmetic of
 * modules/hash_encoder.py:183-205 + modules/utils.py:19-42 (f64 sizes) and of the in-kernel
 * grid_scale/grid_resolution (hash_encoder.py:73-80, f32) and handed to every hash kernel by value, so
 * the CPU oracle and the HIP kernels index with the very same numbers. */
typedef struct ngp_hash_levels {
    int32_t  n_levels;                 /* L, <= NGP_MAX_LEVELS                                  */
    int32_t  n_features;               /* F, features per entry (2)                             */
    int32_t  begin_fast_hash_level;    /* first level that uses the xor-prime hash              */
    int32_t  total_entries;            /* sum of map_size                                       */
    float    scale[NGP_MAX_LEVELS];    /* base*exp(l*log_b)-1, f32                              */
    uint32_t resolution[NGP_MAX_LEVELS]; /* ceil(scale)+1                                       */
    uint32_t map_size[NGP_MAX_LEVELS]; /* entries in level l                                    */
    uint32_t offset[NGP_MAX_LEVELS];   /* first entry of level l                                */
    uint32_t bwd_plan;                 /* NGP_BWD_PLAN_* bits: task plan of the LDS-sliced scatter-add over this table
                                          (ngp_hash_bwd_sliced_*); 0 after ngp_hash_levels_init; every other entry point ignores it */
} ngp_hash_levels;

/* Task-plan modes of the LDS-sliced scatter-add (ngp_hash_levels.bwd_plan).  They are part of the level table the caller hands to
 * EVERY ngp_hash_bwd_sliced_* call of one step (prep, main*, adam_prefix, plan must see the same bits): the library keeps no mode
 * state of its own (until ABI version 1 these were per-thread switches, ngp_hash_bwd_sliced_deterministic / _concentrated).
 * DETERMINISTIC: the default plan replicates the coarse levels over sample ranges whose owners meet in dtable with float atomics
 *   (order-dependent at ~1e-7) and pre-sums equal-cell runs in groups that depend on which wave took which piece of the sample list.
 *   With this bit every slice has ONE owner (no float atomics; with _main_adam the optimizer runs in the flush of EVERY level,
 *   _adam_prefix = 0) and a pre-summing group never spans two pieces: the result is a function of the inputs.  Slower on the coarse
 *   levels; bench.py conditions its model in this mode so that two processes reach the same state.
 * CONCENTRATED: for scenes that fill a small part of the box (multi-cascade scenes) the coarse hashed levels (resolution <= 256) get
 *   sample-range replicas like the dense ones -- a few hot cells otherwise load a handful of slice owners with several times the mean
 *   (C3: the launch 2.2 ms with the XCDs busy 55 % of it; 1.7 ms in this mode).  The levels that get replicas leave the set
 *   ngp_hash_bwd_sliced_main_adam updates in its flush (ngp_hash_bwd_sliced_adam_prefix says where that set starts). */
#define NGP_BWD_PLAN_DETERMINISTIC 1u
#define NGP_BWD_PLAN_CONCENTRATED  2u

int ngp_abi_version(void);

/* Entry points that NO default path of the package calls (earlier rounds' forms kept for A/B runs, the overlapped-exchange experiment, host-side
 * introspection and diagnostics) are declared in ngp_hip_experimental.h; the library exports both sets. */

/* Host-only helper: fills `lv` like HashEncoder.__init__ (hash_encoder.py:183-205). Returns 0. */
int ngp_hash_levels_init(ngp_hash_levels* lv, double max_params, int levels, double base_res,
                         double max_res, int features);

/* ---- a-1  ray_aabb_intersect (modules/intersection.py:8-37) ---------------------------------- */
int ngp_ray_aabb(const float* rays_o, const float* rays_d, float scale, int n_rays,
                 float* hits_t /*[n,2]*/, void* stream);

/* ---- a-2  raymarching_train_kernel (modules/ray_march.py:8-123) ------------------------------
 * The reference's single kernel (count pass, two global atomics per ray, write pass) is split into
 * three deterministic launches:
 *   count : one lane per ray walks the f32 orbit once, stores every emitted (t,dt) into the ray's
 *           private staging row stage[r*max_samples + k] and its sample count into counts[r];
 *   scan  : exclusive prefix sum over counts -> rays_a[r] = (r, start, count) IN RAY ORDER and
 *           total[0] = number of samples (replaces counter[0]/counter[1] atomics, ray_march.py:76-81);
 *   write : sample-parallel expansion of the staged (t,dt) into xyzs/dirs/deltas/ts.            */
/* (hits_t may be NULL in the _ex form below: the slab test of ngp_ray_aabb is then evaluated inline.) */
/* Optional acceleration of `count`: coarse[k] bit = any occupied cell among the 512 Morton codes of 8^3-cell block k
 * (built by ngp_bitfield_coarsen whenever the bitfield changes; cascades*grid^3/512 bits).  Purely a shortcut for
 * provably empty cells -- results are bit-identical with coarse == NULL. */
int ngp_bitfield_coarsen(const uint8_t* density_bitfield, int cascades, int grid_size, uint32_t* coarse,
                         void* stream);
int ngp_march_train_count_ex(const float* rays_o, const float* rays_d, const float* hits_t,
                             const uint8_t* density_bitfield, const uint32_t* coarse, const float* noise,
                             int cascades, int grid_size, float scale, float exp_step_factor, int max_samples,
                             int n_rays, float* stage, int32_t* counts, void* stream);
/* The whole training march in ONE launch (count + allocation + expansion): a 16-wave block marches 32 rays, takes its output
 * range from ctr[0] with one atomic add and writes its rays' samples itself.  Per ray the samples are those of the chain above
 * (bit for bit, contiguous, in march order); the rays follow each other in the order their blocks finished, as in the reference
 * (ray_march.py:76-80: N_eff_samples / counter atomic adds) -- rays_a[r] = (r, start, count) says where.  ctr = [2] int32 scratch,
 * zero before the FIRST launch only (the kernel leaves it zero); total[0] = number of samples.  hits_t / coarse nullable as in
 * ngp_march_train_count_ex; stage = [n_rays * max_samples * 2] scratch. */
int ngp_march_train_fused(const float* rays_o, const float* rays_d, const float* hits_t,
                          const uint8_t* density_bitfield, const uint32_t* coarse, const float* noise, int cascades,
                          int grid_size, float scale, float exp_step_factor, int max_samples, int n_rays,
                          float* stage, int32_t* ctr, int32_t* rays_a, int32_t* total, float* xyzs, float* dirs,
                          float* deltas, float* ts, void* stream);
/* ngp_march_train_fused writes start + k for every emitted sample: xyzs / dirs / deltas / ts must have n_rays * max_samples rows.
 * With smaller arrays use this form: samples at or beyond `capacity` rows are dropped, total[0] still counts them. */
int ngp_march_train_fused_cap(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                              const uint32_t* coarse, const float* noise, int cascades, int grid_size, float scale,
                              float exp_step_factor, int max_samples, int n_rays, long long capacity, float* stage, int32_t* ctr,
                              int32_t* rays_a, int32_t* total, float* xyzs, float* dirs, float* deltas, float* ts, void* stream);
/* The same with the jitter drawn in the kernel: ray r gets rng_uniform(seed, r) (splitmix64 -> 24-bit uniform in [0, 1); the
 * reference draws torch.rand_like, ray_march.py:138).  ngp_rng_uniform writes those values out (tests / callers wanting a tensor). */
int ngp_march_train_fused_rng(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                              const uint32_t* coarse, unsigned long long seed, int cascades, int grid_size, float scale,
                              float exp_step_factor, int max_samples, int n_rays, float* stage, int32_t* ctr, int32_t* rays_a,
                              int32_t* total, float* xyzs, float* dirs, float* deltas, float* ts, void* stream);
/* Round 5: the same launch with its SHAPE chosen by the caller (the side-stream prefetch of FusedTrainer puts the march underneath
 * other kernels of the step): block_waves in {4, 8, 16} waves per block (16 = the entry points above), lds_pad_bytes of dynamic
 * LDS the kernel never touches (caps the blocks one CU takes at a time; 0 = none).  noise == NULL: jitter rng_uniform(seed, r),
 * else the vector (seed ignored).  Per ray the samples are bit for bit those of every other form (ray_march.py:8-123). */
int ngp_march_train_fused_shaped(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* density_bitfield,
                                 const uint32_t* coarse, const float* noise, unsigned long long seed, int cascades, int grid_size,
                                 float scale, float exp_step_factor, int max_samples, int n_rays, int block_waves, int lds_pad_bytes,
                                 float* stage, int32_t* ctr, int32_t* rays_a, int32_t* total, float* xyzs, float* dirs, float* deltas,
                                 float* ts, void* stream);
int ngp_rng_uniform(unsigned long long seed, int n, float* out, void* stream);
int ngp_march_train_scan(const int32_t* counts, int n_rays, int32_t* rays_a /*[n,3]*/,
                         int32_t* total /*[1]*/, void* stream);
int ngp_march_train_write(const float* rays_o, const float* rays_d, const int32_t* rays_a,
                          const float* stage, int max_samples, int n_rays,
                          float* xyzs, float* dirs, float* deltas, float* ts, void* stream);

/* ---- a-3  raymarching_test_kernel (modules/ray_march.py:197-268) ----------------------------- */
int ngp_march_test(const float* rays_o, const float* rays_d, float* hits_t /*in-out*/,
                   const int64_t* alive_indices, const uint8_t* density_bitfield,
                   int cascades, int grid_size, float scale, float exp_step_factor,
                   int max_samples, int n_alive,
                   int64_t* ray_indices, uint8_t* valid_mask, float* deltas, float* ts,
                   int32_t* samples_counter, void* stream);

/* ---- a-4  hash_encoder_kernel fp32 + its autodiff backward (modules/hash_encoder.py:89-143,269) */
int ngp_hash_fwd_f32(const float* xyzs /*[n,3] in [0,1]*/, const float* table,
                     const ngp_hash_levels* lv, int n, float* out /*[n, L*F]*/, void* stream);
/* dtable must be zero-filled by the caller (or hold a gradient to accumulate into). */
int ngp_hash_bwd_f32(const float* xyzs, const float* dout /*[n, L*F]*/,
                     const ngp_hash_levels* lv, int n, float* dtable, void* stream);

/* Sync-free forms used by the fused training step: the sample count is read ON THE DEVICE from n_dev[0] (the
 * `total` written by ngp_march_train_scan; NULL = use n_max), buffers are sized for n_max, and `normalize` fuses the
 * caller-side position normalisation (x - lo) / (hi - lo) of modules/networks.py:144 (same two f32 operations).
 * enc_pairs = 1 (L = 16, F = 2 only) selects the PAIR-MAJOR encoding layout of the fused path:
 *     enc[(p * n_max + i) * 4 + slot],  plane p = min(l, 15 - l),  slot = 2 * (l >= 8) + f
 * i.e. eight [n_max, 4] planes, each written by the workgroups of one XCD (16 contiguous bytes per sample). */
int ngp_hash_fwd_f32_ex(const float* xyzs, const float* table, const ngp_hash_levels* lv, int n_max,
                        const int32_t* n_dev, int normalize, float lo, float hi, int enc_pairs, float* out,
                        void* stream);
/* bf16-stored table (F = 2; entries packed as bf16 pairs): f32 interpolation and f32 output, i.e. bit-identical to
 * ngp_hash_fwd_f32_ex on the bf16-rounded table at half the gather bytes (588 B/sample algorithmic).  The gradient of the
 * encoding w.r.t. the table does not depend on the table values: the backward is ngp_hash_bwd_f32_ex into the fp32 master
 * gradient. */
int ngp_hash_fwd_bf16_ex(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n_max,
                         const int32_t* n_dev, int normalize, float lo, float hi, int enc_pairs, float* out,
                         void* stream);
/* Round 5 -- the same encoder over a LIST of samples (the chunked forward below): rows list[0 .. *n_list) of xyzs are encoded into
 * the same rows of out (16 levels x 2 features; pair-major planes of stride n_max when enc_pairs).  table_kind 0: fp32 table,
 * 1: its bf16 storage copy.  Bit-identical per row to ngp_hash_fwd_f32_ex / _bf16_ex (hash_encoder.py:89-143).  Returns -2 where
 * the specialised kernel does not apply (other table shapes; tables of 4 GB and more): the caller encodes every row instead. */
int ngp_hash_fwd_list(const float* xyzs, const void* table, int table_kind, const ngp_hash_levels* lv, int n_max,
                      const int32_t* n_list, const int32_t* list, int normalize, float lo, float hi, int enc_pairs, float* out,
                      void* stream);
/* found_inf (nullable): set to 1 when a non-finite incoming gradient is seen -- GradScaler's inf/nan check
 * (train.py:199) done where the data passes instead of in an extra pass over the 45.7 MB gradient. */

/* ---- a-5  half2 encoder fwd / explicit bwd (modules/hash_encoder_half.py:112-161,164-213) ----
 * table/out/dout/dtable are IEEE binary16 pairs (uint16_t storage). */
int ngp_hash_fwd_f16(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n,
                     uint16_t* out /*[n, L, 2]*/, void* stream);
int ngp_hash_bwd_f16(const float* xyzs, const uint16_t* dout, const ngp_hash_levels* lv, int n,
                     uint16_t* dtable /*[entries,2] f16*/, void* stream);

/* Fused-path forms of the half2 encoder (same arithmetic, the buffers of the fp32 fused path): the forward widens its f16
 * result to f32 (exact) into the natural [n,32] or the pair-major layout; the backward takes the fp32 d_enc of the fused MLP
 * backward, rounds it to f16 like the reference's fp16 output gradient, merges equal-cell runs of consecutive samples before
 * the packed f16 atomic (so the f16 accumulation order differs from a serial run: tolerance-checked), and flags non-finite
 * incoming gradients.  ngp_check_finite_f16 scans the ACCUMULATED f16 gradient (n % 8 == 0) for inf / nan. */
int ngp_hash_fwd_f16_ex(const float* xyzs, const uint16_t* table, const ngp_hash_levels* lv, int n_max,
                        const int32_t* n_dev, int normalize, float lo, float hi, int enc_pairs, float* out,
                        void* stream);
int ngp_check_finite_f16(const uint16_t* g, long long n, int32_t* found_inf, void* stream);

/* ---- a-6  dir_encoder (modules/spherical_harmonics.py:7-42) + analytic backward -------------- */
int ngp_sh16_fwd(const float* dirs /*[n,3]*/, int n, float* out /*[n,16]*/, void* stream);
int ngp_sh16_bwd(const float* dirs, const float* dout /*[n,16]*/, int n, float* ddirs /*[n,3]*/,
                 void* stream);

/* ---- a-7  volume_rendering_kernel + closed-form backward (modules/volume_train.py:6-48,160) ---
 * rgbs_is_half selects fp16 (autocast) or fp32 `rgbs`; the gradient is written in the same dtype.
 * Outputs are indexed by ray_idx = rays_a[n,0] like the reference. dL_dopacity / dL_ddepth / dL_dws may
 * be NULL (== zeros); opacity/depth/rgb/ws are the forward outputs. */
int ngp_composite_train_fwd(const float* sigmas, const void* rgbs, int rgbs_is_half,
                            const float* deltas, const float* ts, const int32_t* rays_a,
                            float T_threshold, int n_rays,
                            int32_t* total_samples /*[n]*/, float* opacity, float* depth,
                            float* rgb /*[n,3]*/, float* ws /*[S]*/, void* stream);
int ngp_composite_train_bwd(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb,
                            const float* dL_dws, const float* sigmas, const void* rgbs,
                            int rgbs_is_half, const float* deltas, const float* ts,
                            const int32_t* rays_a, const float* opacity, const float* depth,
                            const float* rgb, const float* ws, float T_threshold, int n_rays,
                            float* dL_dsigmas, void* dL_drgbs, void* stream);
/* ... with the background blend of rendering.py:219-226 (results['rgb'] = rgb + rgb_bg * (1 - opacity), rgb_bg = ones behind
 * synthetic scenes) inside the two launches: _fwd_bg also writes rgb_out[n,3] = rgb + bg (1 - opacity) (nullable; `rgb` stays the
 * unblended colour the backward reads); _bwd_bg takes dL_drgb as the gradient of the BLENDED colour and adds the blend's share,
 * -bg sum_c dL_drgb, to d opacity.  bg == 0: the two entries above. */
int ngp_composite_train_fwd_bg(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                               const int32_t* rays_a, float T_threshold, int n_rays, int32_t* total_samples, float* opacity,
                               float* depth, float* rgb, float* ws, float* rgb_out, float bg, void* stream);
int ngp_composite_train_bwd_bg(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                               const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas, const float* ts,
                               const int32_t* rays_a, const float* opacity, const float* depth, const float* rgb, const float* ws,
                               float T_threshold, int n_rays, float* dL_dsigmas, void* dL_drgbs, float bg, void* stream);

/* Round 5 -- chunked forward: do not shade what compositing will never read.  The reference shades every marched sample and then
 * ignores those behind T <= T_threshold (volume_train.py:38); its evaluation loop (rendering.py:62-158) already shades in rounds
 * and drops finished rays.  One call per round: samples [begin, begin + len) of every ray still alive are appended to `list`
 * (count[0] += their number; count_zero, nullable, is set to 0 for a later round's counter).  begin > 0: the ray's transmittance
 * state T_state[row of rays_a] (n_rays floats) is first advanced over [prev_begin, begin) -- which must have been shaded -- as
 * T *= exp(-sum sigma delta), and the ray is retired when T <= thr_stop; begin == 0 initialises the state.  begin / len / prev_begin
 * must be multiples of 64 and thr_stop at most half the compositing threshold: ngp_composite_train_* read a ray 64 samples at a
 * time and skip a group when T <= threshold at its start, so every group they read has then been shaded.  Per ray rgb / opacity /
 * depth / ws / vr and every gradient are those of shading everything (tests/test_gpu_chunked.py). */
int ngp_chunk_schedule(const int32_t* rays_a, const float* sigmas, const float* deltas, int n_rays, int begin, int len,
                       int prev_begin, float thr_stop, float* T_state, int32_t* list, int32_t* count, int32_t* count_zero,
                       void* stream);
/* The list of ngp_live_compact in block-completion order (each ray's samples contiguous; the *_live kernels do not depend on the
 * order) from one launch with one atomic per 64 rays: live_total[0] must be 0 at launch, live_zero (nullable) is cleared. */
int ngp_live_list(const int32_t* rays_a, const int32_t* vr_per_ray /*[n], by ray index*/, int n_rays, int32_t* live_idx,
                  int32_t* live_total, int32_t* live_zero, void* stream);
/* The same launch, and the compacted live-sample list of ngp_live_compact as a by-product: every 16-ray block appends its rays'
 * first vr[r] samples to live_idx at an offset it takes from live_total with ONE atomic add, so the list is in block-completion
 * order (each ray contiguous) -- the *_live kernels do not depend on the order.  live_total[0] must be 0 at launch; live_zero
 * (nullable) is set to 0 by the kernel: a caller alternating two counters gets the next step's counter cleared for free.
 * live_idx == NULL: plain ngp_composite_train_fused. */
int ngp_composite_train_fused_live(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas,
                                   const float* ts, const int32_t* rays_a, const float* target, float bg,
                                   const float* loss_scale, float T_threshold, int n_rays, int32_t* vr_per_ray,
                                   float* opacity, float* depth, float* rgb, float* ws, float* d_sigmas, void* d_rgbs,
                                   float* sq_err, int32_t* live_idx, int32_t* live_total, int32_t* live_zero,
                                   void* stream);

/* Live-sample list for the backward pass: the first vr_per_ray[r] samples of ray r (those in front of the early-termination
 * point, volume_train.py:31-47) are the only ones with a non-zero gradient.  live_idx[j] = sample index of the j-th live
 * sample (ray order), live_total[0] = their number, live_off = [n_rays] scratch.  ngp_mlp_bwd_live / ngp_hash_bwd_*_live run
 * on that list: position j reads sample live_idx[j] (enc, dirs, gradients, xyzs) and d_enc is written / read at position j. */
int ngp_live_compact(const int32_t* rays_a, const int32_t* vr_per_ray /*[n], by ray index*/, int n_rays, int32_t* live_off,
                     int32_t* live_idx, int32_t* live_total, void* stream);
int ngp_mlp_bwd_live(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas, const uint16_t* drgbs,
                     int n_max, const int32_t* n_dev, const int32_t* live_idx, int enc_pairs, float* d_enc, float* dW,
                     int32_t* found_inf, void* stream);
/* The same backward with the weight gradients left as PER-BLOCK SLABS instead of float atomics on dW: block b of the launch
 * writes its sums of all 9408 weights to dW_parts[b * 9408 ...] with plain stores (256 blocks x 9408 same-address atomics were
 * 13 us of a 72 us launch).  dW_parts: ngp_mlp_dw_parts_max() * 9408 floats, no initialisation needed.  Returns the number of
 * slabs written (>= 1; 0 for n_max <= 0; < 0 on error).  ngp_mlp_dw_reduce ADDS the slabs to dW [9408] -- or the trainer's
 * ngp_train_prologue_reduce does, in the launch it needs anyway. */
int ngp_mlp_dw_parts_max(void);
int ngp_mlp_bwd_live_parts(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas,
                           const uint16_t* drgbs, int n_max, const int32_t* n_dev, const int32_t* live_idx, int enc_pairs,
                           float* d_enc, float* dW_parts, int32_t* found_inf, void* stream);
int ngp_mlp_dw_reduce(const float* dW_parts, int n_parts, float* dW, void* stream);
int ngp_hash_bwd_f32_live(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                          const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, float* dtable,
                          int32_t* found_inf, void* stream);
int ngp_hash_bwd_f16_live(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                          const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, uint16_t* dtable,
                          int32_t* found_inf, void* stream);
/* The same scatter-add (autodiff backward of modules/hash_encoder.py:89-143, call site :269; F = 2) WITHOUT global float
 * atomics: the gradient table is cut into slices of 8192 entries (128 KB of f64 pairs), each owned by one workgroup that
 * accumulates in its LDS and adds the slice to the table once (csrc/hash_bwd_lds.hip).  Same arguments and semantics as
 * ngp_hash_bwd_f32_live (dtable is accumulated into) plus a caller-owned scratch buffer of
 * ngp_hash_bwd_sliced_workspace(lv, n_max) bytes: compact positions + one hit BIT per (level, slice, sample) + the queue heads
 * of the persistent workgroups.  Returns -2 when the level table does not fit the formulation (F != 2, a level of more than
 * 64 slices = 2^19 entries, an odd level size): the caller then uses ngp_hash_bwd_f32_live.
 * The workspace belongs to ONE call sequence at a time: prep -> main of one backward must not be interleaved with another
 * backward's on another stream using the same buffer (the bitmaps and the queue heads live in it). */
long long ngp_hash_bwd_sliced_workspace(const ngp_hash_levels* lv, int n_max);
/* The two halves as separate entry points: the prepass (compact positions + hit bitmaps) needs only positions and live list, so it
 * can be issued before the MLP backward that produces `dout`; `main` consumes the workspace it filled. */
int ngp_hash_bwd_sliced_prep(const float* xyzs, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                             const int32_t* live_idx, int normalize, float lo, float hi, void* workspace,
                             long long workspace_bytes, void* stream);
int ngp_hash_bwd_sliced_main(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                             float* dtable, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                             void* stream);
/* ngp_hash_bwd_sliced_main / _main_f16 (table_is_f16) whose first workgroups ALSO finish the MLP backward: dW[9408] += the sum of
 * the n_parts weight-gradient slabs ngp_mlp_bwd_live_parts left (= ngp_mlp_dw_reduce, without a launch of its own: in a 200 us
 * launch of persistent workgroups the ~2 us are invisible; the trainer's single-GPU step). */
int ngp_hash_bwd_sliced_main_slabs(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                   void* dtable, int table_is_f16, int32_t* found_inf, const void* workspace,
                                   long long workspace_bytes, const float* mlp_dw_parts, int n_parts, float* mlp_dw, void* stream);
/* Round 5 -- scatter-add WITH the optimizer (replaces train.py:197-201 on the path modules/hash_encoder.py:269 feeds; one GPU).
 * The owner of a slice that is not replicated over sample ranges (every level of 64 slices = 2^19 entries: the hashed levels)
 * holds that slice's complete gradient in LDS when its task ends; instead of adding it to dtable for ngp_adam_all_ex to read
 * back, it applies the same torch.optim.Adam update (eps as given, unscale by state_f[1], skip when state_i[4]) to EVERY entry of
 * its slice of table / table_m / table_v and refreshes table_bf16 (nullable): 24 B per parameter instead of 40.  The remaining
 * (coarse, replicated) levels still accumulate into dtable; they occupy table floats [0, ngp_hash_bwd_sliced_adam_prefix(lv)),
 * and the caller runs ngp_adam_all_ex over exactly that range afterwards.  Results are bit-identical to
 * ngp_hash_bwd_sliced_main + ngp_adam_all_ex over the whole table (tests/test_gpu_flush_adam.py).
 * state_f / state_i must hold THIS step's decision already: run ngp_train_prologue before this call; the inf flag it consumes comes
 * from ngp_mlp_bwd_live[_parts], which checks the d_enc values this kernel reads (this entry point does not write found_inf).
 * mlp_dw_parts / n_parts / mlp_dw: the optional slab sum of ngp_hash_bwd_sliced_main_slabs (NULL / 0: none).
 * _adam_prefix returns -2 (and _main_adam -2) when the level table has no non-replicated levels. */
long long ngp_hash_bwd_sliced_adam_prefix(const ngp_hash_levels* lv);
/* The same launch with the step's scalar bookkeeping inside it (ngp_train_prologue's arguments; train.py:197-201): every workgroup
 * evaluates the GradScaler / schedule decision on a private copy of state_f / state_i when it starts (as the previous step left
 * them + this step's inf flag from the MLP backward), its flushes use that copy, and the last workgroup out stores it -- after the
 * launch state_f / state_i are exactly what ngp_train_prologue would have left, and no one-thread launch sits in front of the
 * scatter-add.  Bit-identical to ngp_train_prologue + ngp_hash_bwd_sliced_main_adam (tests/test_gpu_flush_adam.py). */
int ngp_hash_bwd_sliced_main_adam_step(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                       float* dtable, const void* workspace, long long workspace_bytes, const float* mlp_dw_parts,
                                       int n_parts, float* mlp_dw, float* table, float* table_m, float* table_v, uint16_t* table_bf16,
                                       float* state_f, int32_t* state_i, float lr0, float eta_min, int t_max, float beta1, float beta2,
                                       float eps, float growth, float backoff, int growth_interval, void* stream);
/* the half2 encoder's backward (hash_encoder_half.py:163-213) over the same prepass: dtable_f16 = fp16 pairs [entries][2]; the
 * encoder's fp16 arithmetic per contribution (cell cast to f16, w * g rounded to f16), the owner's f64 sum rounded to fp16 once */
int ngp_hash_bwd_sliced_main_f16(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                 uint16_t* dtable_f16, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                                 void* stream);
int ngp_hash_bwd_f32_sliced(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev,
                            const int32_t* live_idx, int normalize, float lo, float hi, int enc_pairs, float* dtable,
                            int32_t* found_inf, void* workspace, long long workspace_bytes, void* stream);

/* ---- a-8  composite_test (modules/volume_render_test.py:4-54) -------------------------------- */
int ngp_composite_test(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas,
                       const float* ts, const int64_t* pack_info /*[n,2]*/,
                       int64_t* alive_indices /*in-out*/, float T_threshold, int n_alive,
                       float* opacity, float* depth, float* rgb, void* stream);

/* ---- a-9  the two MLPs + TruncExp + direction glue, fused (modules/networks.py:18-30,111-132,136-166,293-380
 * under torch.autocast(fp16), train.py:177).  Default architecture only: xyz_encoder 32->64->16, rgb_net
 * [SH16|h16]->64->64->3, bias-free.  Weights are the fp32 master tensors in nn.Linear layout [out][in];
 * ngp_mlp_pack rounds them to fp16 and lays them out as MFMA fragments (ngp_mlp_wpack_halfs() uint16 elements).
 *   fwd : enc [n,32] f32 (hash-grid output), dirs [n,3] f32 (raw ray directions; normalisation, (d+1)/2 and SH16
 *         happen inside) -> sigmas [n] f32, rgbs [n,3] f16.  dirs == NULL or rgbs == NULL: density only.
 *   bwd : recomputes the forward, consumes dL/dsigmas [n] f32 and dL/drgbs [n,3] f16, writes dL/denc [n,32] f32 and
 *         ACCUMULATES the weight gradients into dW [9408] f32 = W1|W2|W3|W4|W5 row-major (caller zero-fills). */
int ngp_mlp_wpack_halfs(void);
int ngp_mlp_pack(const float* W1, const float* W2, const float* W3, const float* W4, const float* W5,
                 int enc_pairs, uint16_t* wpack, void* stream);      /* enc_pairs: which enc layout the image is for */
int ngp_mlp_fwd(const float* enc, const float* dirs, const uint16_t* wpack, int n, float* sigmas,
                uint16_t* rgbs, void* stream);
int ngp_mlp_bwd(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas,
                const uint16_t* drgbs, int n, float* d_enc, float* dW, void* stream);

int ngp_mlp_fwd_ex(const float* enc, const float* dirs, const uint16_t* wpack, int n_max, const int32_t* n_dev,
                   int enc_pairs, float* sigmas, uint16_t* rgbs, void* stream);
/* Round 5 -- the forward over a LIST of samples: position j < *n_list shades sample list[j] (reads its enc row and direction, writes
 * its sigma / rgb; networks.py:136-166).  Bit-identical per sample to ngp_mlp_fwd_ex. */
int ngp_mlp_fwd_list(const float* enc, const float* dirs, const uint16_t* wpack, int n_max, const int32_t* n_list,
                     const int32_t* list, int enc_pairs, float* sigmas, uint16_t* rgbs, void* stream);

/* Stream plumbing for a caller that runs the march of the NEXT batch on a second stream (FusedTrainer): events that order two
 * streams of THIS device without the system-scope cache write-back + invalidate a default HIP event performs when it is recorded
 * (hipEventDisableTiming | hipEventDisableSystemFence).  Not for host-visible results. */
int ngp_event_create(void** event);
int ngp_event_record(void* event, void* stream);
int ngp_stream_wait_event(void* stream, void* event);
int ngp_event_destroy(void* event);
/* ... and the host-side wait for one (round 6): returns when the work recorded in front of the event has completed.  FusedTrainer can wait
 * this way (NGP_EXPERIMENT prefetch_host_wait=1) for a prefetched march that was issued a whole step earlier: a stream-side wait for an
 * event of another queue costs the waiting stream ~10 us even when the event completed long ago. */
int ngp_event_synchronize(void* event);
/* A non-blocking stream of the lowest priority the device offers (for the prefetched march: it should take only the CUs the step's
 * own kernels leave free); *least / *greatest (nullable) = hipDeviceGetStreamPriorityRange. */
int ngp_stream_create_low_priority(void** stream, int* least, int* greatest);
int ngp_stream_destroy(void* stream);
/* Pinned host memory + an asynchronous device-to-host copy on a stream of the caller's choice: a word the device reports into
 * without anybody waiting for it (FusedTrainer: the sample count of each prefetched march, read by later steps to place the next
 * one).  The caller frees the memory after synchronising the streams it copied on. */
int ngp_host_alloc(void** host, long long bytes);
int ngp_host_free(void* host);
int ngp_copy_to_host_async(void* host, const void* dev, long long bytes, void* stream);

/* ---- the fused training render as ONE entry per direction (round 5; replaces the launch sequence of reference
 * modules/rendering.py:161-228 + modules/networks.py:152-166 and their autograd backward for one batch of rays).  The caller keeps
 * one argument block per sample arena and patches the per-call pointers; every buffer is caller-owned device memory.
 * _fwd: [coarse occupancy table when rebuild_coarse] -> ngp_march_train_fused -> ngp_hash_fwd_{f32,bf16,f16}_ex -> ngp_mlp_pack ->
 *       ngp_mlp_fwd_ex -> ngp_composite_train_fwd.
 * _bwd: ngp_composite_train_bwd -> ngp_live_compact -> ngp_mlp_bwd_live -> ngp_hash_bwd_f32_sliced into dW / dtable, which the caller
 *       hands over CLEARED (or the
 *       float-atomic kernel when the level table does not fit it / force_atomic; table_kind 2: the half2 encoder's fp16 forms).
 * Same kernels, same results as the individual entry points in that order (tests/test_gpu_fused.py). */
typedef struct ngp_render_args {
    const float* rays_o; const float* rays_d; const float* hits_t /*nullable: slab test inline*/; const float* noise /*[n_rays] jitter*/;
    int32_t n_rays, max_samples;
    const uint8_t* bitfield; uint32_t* coarse; int32_t rebuild_coarse, cascades, grid_size;
    float scale, exp_step_factor, T_threshold;
    const void* table; int32_t table_kind /*0 fp32, 1 bf16 copy, 2 f16 copy (half2 encoder)*/, enc_pairs;
    const ngp_hash_levels* levels; float lo, hi;
    const float* w[5]; uint16_t* wpack;
    long long cap;                                   /* rows of the per-sample buffers below (n_rays * max_samples) */
    float* stage; int32_t* march_ctr; float* xyzs; float* dirs; float* deltas; float* ts; float* enc; float* sigmas; uint16_t* rgbs; float* ws;
    int32_t* rays_a; int32_t* total; int32_t* vr_per_ray; float* opacity; float* depth; float* rgb;          /* per-ray outputs */
    const float* g_opacity; const float* g_depth; const float* g_rgb; const float* g_ws;                    /* backward: nullable but g_rgb */
    float* d_sigmas; uint16_t* d_rgbs; float* d_enc; int32_t* live_off; int32_t* live_idx; int32_t* live_total;
    void* workspace; long long workspace_bytes;      /* ngp_hash_bwd_sliced_workspace(levels, cap) bytes */
    int32_t force_atomic, reserved;
    /* round 6 (both nullable: NULL = round 5's launches).  dW_parts: ngp_mlp_dw_parts_max() * 9408 floats -- the MLP backward leaves its
     * weight gradients as per-block slabs and the head of the scatter-add launch adds them to dW (ngp_hash_bwd_sliced_main_slabs) instead of
     * 256 x 9408 same-address float atomics.  live_zero: a second counter that is 0 at launch; the live list is then built by ngp_live_list
     * (one atomic per 64 rays, block-completion order) into live_total -- WHICH MUST BE 0 AT LAUNCH -- and live_zero is what the launch clears
     * for the next call (a caller alternating the two never clears one itself). */
    float* dW_parts; int32_t* live_zero;
    float* dW /*[9408], cleared by the caller*/; void* dtable /*f32 [entries * 2]; table_kind 2: f16; cleared by the caller*/; long long dtable_bytes;
    /* ABI 3.  bg: the background the composited colour is blended over (rendering.py:219-226: 1 behind synthetic scenes, 0 behind real
     * ones).  rgb_out (nullable, [n_rays,3]): _fwd also writes rgb + bg (1 - opacity) there -- what the loss sees; `rgb` stays the
     * unblended colour the backward reads.  _bwd takes g_rgb as the gradient of the BLENDED colour when bg != 0 (adds -bg sum_c g_rgb to
     * d opacity).  clear_grads != 0: _bwd clears dW and dtable itself (one fill launch behind the compositing backward; dtable_bytes % 16
     * == 0) instead of expecting them cleared. */
    float bg; int32_t clear_grads; float* rgb_out;
} ngp_render_args;
int ngp_render_train_fwd(const ngp_render_args* args, void* stream);
int ngp_render_train_bwd(const ngp_render_args* args, void* stream);

/* ---- f-2  device-resident optimisation-step epilogue (reference train.py:193-201: mse_loss, GradScaler,
 * Adam(eps=1e-15), CosineAnnealingLR, zero_grad) -- see csrc/optim.hip.
 * state_f[8] f32: [0] loss scale, [1] 1/scale of this step, [2] lr, [3] 1-beta1^t, [4] sqrt(1-beta2^t), [5] last loss
 * state_i[8] i32: [0] iteration, [1] optimizer steps taken, [2] growth tracker, [3] found_inf (set by the backward
 *                 kernels, consumed + cleared by the prologue), [4] skip flag of this step, [5] skipped steps so far */
int ngp_mse_loss_grad(const float* rgb, const float* opacity, const float* target, float bg, int n_rays,
                      float* state_f, float* g_rgb, float* g_opacity, void* stream);
/* ... per ray, from as many blocks as the batch needs: the same g_rgb / g_opacity (loss-scaled by state_f[0]) and, instead of the
 * reduced loss, the per-ray squared error sq_err[r] (nullable) for whoever logs it (train.py:193). */
int ngp_mse_loss_grad_rays(const float* rgb, const float* opacity, const float* target, float bg, int n_rays,
                           const float* state_f, float* g_rgb, float* g_opacity, float* sq_err, void* stream);
int ngp_train_prologue(float* state_f, int32_t* state_i, float lr0, float eta_min, int t_max, float beta1,
                       float beta2, float growth, float backoff, int growth_interval, void* stream);
/* The prologue and, in the same launch, ngp_mlp_dw_reduce(dW_parts, n_parts, dW). */
int ngp_train_prologue_reduce(float* state_f, int32_t* state_i, float lr0, float eta_min, int t_max, float beta1,
                              float beta2, float growth, float backoff, int growth_interval, const float* dW_parts,
                              int n_parts, float* dW, void* stream);
/* Prologue for an optimizer driven by torch.cuda.amp.GradScaler (train.py:143-149,198-201 with apex.optimizers.FusedAdam =
 * compat/apex): grad_scale / found_inf are the scaler's device tensors (NULL: scale 1 / never skip), lr the host-side scheduler's
 * value; fills state_f[1..4] and state_i[1,4] for ngp_adam_step; the bias-correction step count advances on non-skipped steps. */
int ngp_adam_amp_prologue(float* state_f, int32_t* state_i, const float* grad_scale, const float* found_inf, float lr,
                          float beta1, float beta2, void* stream);
/* p, g, m, v: n floats each (n % 4 == 0, 16-byte aligned); g is unscaled on the fly and zero-filled. */
/* The same pass over up to NGP_ADAM_MULTI_MAX tensors in ONE launch (host arrays of device pointers and element counts, read at
 * call time): what an optimizer over model.parameters() -- the table and the five weight matrices, train.py:143-149 -- needs per
 * step instead of one launch per tensor. */
#define NGP_ADAM_MULTI_MAX 16
int ngp_adam_multi(int n_tensors, float* const* p, float* const* g, float* const* m, float* const* v, const long long* n,
                   const float* state_f, const int32_t* state_i, float beta1, float beta2, float eps, void* stream);
/* GradScaler's inf / nan check (train.py:199, torch.amp.GradScaler._check_inf_per_device) over the same tensor list, read-only:
 * found_inf[0] (a float32 device scalar, cleared by the caller) becomes 1.0f when any gradient value is not finite. */
int ngp_check_finite_multi(int n_tensors, const float* const* g, const long long* n, float* found_inf, void* stream);
/* ... and ngp_adam_amp_prologue in the same launch (one parameter group of <= NGP_ADAM_MULTI_MAX tensors): the last block out runs the
 * prologue on the finished flag.  found_inf must be 0 at launch; clear_next (nullable) is a second flag this launch clears for the
 * caller's NEXT step -- GradScaler.update() (train.py:200) reads found_inf after step() returns, so a caller alternating two flags never
 * issues a fill.  done: 17 x 32 uint32 of device memory (arrival counters 128 bytes apart), 0 between launches. */
int ngp_adam_amp_check_prologue(int n_tensors, const float* const* g, const long long* n, float* found_inf, float* clear_next,
                                uint32_t* done, float* state_f, int32_t* state_i, const float* grad_scale, float lr, float beta1,
                                float beta2, void* stream);
/* dst[i] = bf16(src[i]), round-to-nearest-even; n % 4 == 0 (the per-forward cast of hash_encoder_half.py:367, in bf16). */
int ngp_cast_f32_bf16(const float* src, uint16_t* dst, long long n, void* stream);
/* General form: table_g is fp32 or (grad_is_f16) the half2 encoder's f16 gradient buffer, widened to fp32 on the fly;
 * copy_kind 0 = no storage copy, 1 = bf16, 2 = f16 (the table the f16 forward gathers from, hash_encoder_half.py:367). */
int ngp_adam_all_ex(float* table, void* table_g, int grad_is_f16, float* table_m, float* table_v, long long n,
                    uint16_t* table_16, int copy_kind, float* mlp, float* mlp_g, float* mlp_m, float* mlp_v,
                    const float* state_f, const int32_t* state_i, float beta1, float beta2, float eps, int enc_pairs,
                    uint16_t* wpack, void* stream);
/* Adam on the 9 408 flat MLP weights (W1|W2|W3|W4|W5) + the fp16 fragment repack for the next step, one launch. */
int ngp_adam_mlp_pack(float* p, float* g, float* m, float* v, const float* state_f, const int32_t* state_i,
                      float beta1, float beta2, float eps, int enc_pairs, uint16_t* wpack, void* stream);

/* ---- a-10 morton3D / morton3D_invert / packbits (modules/utils.py:120-169) -------------------- */
int ngp_morton3d(const int32_t* coords /*[m,3]*/, int m, int32_t* indices, void* stream);
int ngp_morton3d_invert(const int32_t* indices, int m, int32_t* coords /*[m,3]*/, void* stream);
int ngp_packbits(const float* density_grid /*[8k]*/, float threshold, int n_bytes,
                 uint8_t* bitfield, void* stream);

/* ---- f-1  distortion loss (modules/distortion.py:15-119): per-ray loss[N] (indexed by ray_idx) and the inclusive
 * scans the backward needs; backward writes dL/dws for the live samples addressed by rays_a. */
int ngp_distortion_fwd(const float* ws, const float* deltas, const float* ts, const int32_t* rays_a, int n_rays,
                       float* loss, float* ws_inc, float* wts_inc, void* stream);
int ngp_distortion_bwd(const float* dL_dloss, const float* ws, const float* deltas, const float* ts,
                       const float* ws_inc, const float* wts_inc, const int32_t* rays_a, int n_rays,
                       float* dL_dws, void* stream);

/* ---- f-4  camera rays and training-batch sampling (datasets/ray_utils.py:51-80, datasets/base.py:34-61, train.py:171-184)
 * get_rays:    rays_d[k,i] = sum_j directions[k,j] * c2w[i,j], rays_o[k] = c2w[:,3]; poses = [n,3,4] (per_ray_pose = 1, the
 *              training batch) or one [3,4] pose for all rays (per_ray_pose = 0, an evaluation image).
 * sample_rays: for sample k: image i = img_idx[k] (img0 when img_idx is NULL = 'same_image' sampling), pixel p = pix_idx[k];
 *              pose = poses[i], direction = directions[p], rgb[k] = rays[i, p, 0:3] (rays = [n_img, hw, ray_c] f32, rgb nullable). */
int ngp_get_rays(const float* directions /*[n,3]*/, const float* poses, int per_ray_pose, int n, float* rays_o,
                 float* rays_d, void* stream);
/* up to three n-float copies in one launch (src NULL = slot unused): stages a batch into the static buffers of a captured step */
int ngp_stage_batch(const float* a, float* da, const float* b, float* db, const float* c, float* dc, int n, void* stream);
int ngp_sample_rays(const float* poses /*[n_img,3,4]*/, const float* directions /*[hw,3]*/, const float* rays, int ray_c,
                    long long hw, const int64_t* img_idx, long long img0, const int64_t* pix_idx, int n, float* rays_o,
                    float* rays_d, float* rgb, void* stream);

/* ---- f-3  occupancy-grid update without host round trips (modules/networks.py:181-209,255-290).
 * compact : list[0..count) = cells of ONE cascade with density > threshold, in cell order (deterministic); count is written
 * sample  : m uniform cells (u_cell [m] in [0,1) -> Morton code) + m picks from the list (u_pick [m]) -> Morton indices [2m] and jittered
 *           world positions [2m,3] (u_jit [2m,3]); s = min(2^(c-1), scale), half_grid = s / grid_size
 * all_cells: warm-up variant, cell i = Morton code i
 * scatter : tmp[indices[i]] = sigmas[i] (indices == NULL: identity); a cell drawn twice keeps whichever write lands last, like the
 *           reference's fresh[c, indices] = density.  scatter_max: the largest of them wins (the same one on every run; sigmas > 0)
 * merge   : grid = grid < 0 ? grid : max(grid*decay, tmp); the sum and count of the positive cells leave as per-block partials in
 *           stats[2..] (stats: ngp_occ_stats_floats() floats, no initialisation needed; round 5: no float atomics -- the mean is the
 *           occupancy threshold and must not depend on the order blocks finish in)
 * pack    : every block adds the partials up in one fixed order (stats[0] = sum, stats[1] = count are written for the caller);
 *           bitfield bit = grid > min(stats[0]/stats[1], density_threshold).  n_bytes must be the merge's n / 8. */
int ngp_occ_compact(const float* density_grid, float threshold, int n_cells, int32_t* list, int32_t* count,
                    int32_t* scratch /*[1024]*/, void* stream);
/* m ascending U(0,1) values per set without sorting (normalised partial sums of m + 1 unit exponentials: the order statistics
 * of m iid uniforms, exact in distribution); u: sets * rows * 1024 uniforms (rows = (m + 1 + 1023) / 1024), work: sets * (rows *
 * 1024 + rows) floats, out: [sets][m].  Feeds ngp_occ_sample: ascending uniforms -> ascending cells -> coherent encoder queries. */
int ngp_sorted_uniforms(const float* u, int m, int sets, float* work, float* out, void* stream);
int ngp_occ_sample(const float* u_cell, const float* u_pick, const float* u_jit, const int32_t* list, const int32_t* count,
                   int m, int grid_size, float s, float half_grid, int32_t* indices, float* xyzs, void* stream);
int ngp_occ_all_cells(const float* u_jit, int n_cells, int grid_size, float s, float half_grid, float* xyzs, void* stream);
int ngp_occ_scatter(const int32_t* indices, const float* sigmas, int n, float* tmp, void* stream);
int ngp_occ_scatter_max(const int32_t* indices, const float* sigmas, int n, float* tmp, void* stream);
int ngp_occ_stats_floats(void);
int ngp_occ_merge(float* density_grid, const float* tmp, float decay, int n, float* stats, void* stream);
int ngp_occ_pack(const float* density_grid, float* stats, float density_threshold, int n_bytes, uint8_t* bitfield,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
