/* ngp_hip_experimental.h -- entry points of libngp_hip.so that NO default path of the package calls.
 *
 * Round 6: include/ngp_hip.h is the drop-in boundary -- what modules/, compat/ and FusedTrainer launch with every switch at
 * its default.  What lives here is exported by the same library and typed by the same ctypes table (ngp_hip/lib.py: EXPERIMENTAL), but is reached
 * only through a non-default switch, a test's A/B, or a diagnostic: earlier rounds' forms of a kernel launch that a later form superseded, the
 * per-level-group scatter-add of the overlapped gradient exchange (NGP_COMM_OVERLAP=1, default off until a multi-GPU run has decided), host-side
 * introspection of the scatter-add's task plan, and the per-task timeline buffer.  Nothing here is needed to drive the reference's train.py. */
#ifndef NGP_HIP_EXPERIMENTAL_H
#define NGP_HIP_EXPERIMENTAL_H
#include "ngp_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The whole optimizer pass in one launch: ngp_adam_step (or ngp_adam_step_bf16 when table_bf16 != NULL) on the table and
 * ngp_adam_mlp_pack on the MLP weights, the latter in the first workgroup while the others stream the table. */
int ngp_adam_all(float* table, float* table_g, float* table_m, float* table_v, long long n, uint16_t* table_bf16,
                 float* mlp, float* mlp_g, float* mlp_m, float* mlp_v, const float* state_f, const int32_t* state_i,
                 float beta1, float beta2, float eps, int enc_pairs, uint16_t* wpack, void* stream);

/* One tensor's Adam pass (what ngp_adam_multi does per tensor); p, g, m, v: n floats each (n % 4 == 0, 16-byte aligned); g is unscaled by
 * state_f[1] and cleared.  Superseded on every default path by ngp_adam_all_ex / ngp_adam_multi. */
int ngp_adam_step(float* p, float* g, float* m, float* v, long long n, const float* state_f,
                  const int32_t* state_i, float beta1, float beta2, float eps, void* stream);

/* Same pass, additionally refreshing p_bf16 (n bf16, round-to-nearest-even) -- the table ngp_hash_fwd_bf16_ex gathers from.
 * BASELINE config 2 names a bf16 hash grid; the reference itself has fp32 (hash_encoder.py) and fp16 (hash_encoder_half.py)
 * tables only, so the semantics here are "fp32 master + 16-bit storage copy", as hash_encoder_half.py:367 does for fp16. */
int ngp_adam_step_bf16(float* p, float* g, float* m, float* v, long long n, const float* state_f,
                       const int32_t* state_i, float beta1, float beta2, float eps, uint16_t* p_bf16, void* stream);

/* Trainer fusion of composite forward + MSE gradient (train.py:193, white/black background blend of
 * rendering.py:219-226) + composite backward: one wave per ray, two passes.  loss_scale points at state_f[0]; the
 * per-ray squared error sum_c (rgb_final - target)^2 is written to sq_err[ray] (nullable) for logging. */
int ngp_composite_train_fused(const float* sigmas, const void* rgbs, int rgbs_is_half, const float* deltas,
                              const float* ts, const int32_t* rays_a, const float* target, float bg,
                              const float* loss_scale, float T_threshold, int n_rays, int32_t* vr_per_ray,
                              float* opacity, float* depth, float* rgb, float* ws, float* d_sigmas, void* d_rgbs,
                              float* sq_err, void* stream);

/* The half2 encoder's float-atomic scatter-add over ALL samples (no live list): superseded by ngp_hash_bwd_f16_live and the LDS-sliced form. */
int ngp_hash_bwd_f16_ex(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max,
                        const int32_t* n_dev, int normalize, float lo, float hi, int enc_pairs, uint16_t* dtable,
                        int32_t* found_inf, void* stream);

/* The fp32 float-atomic scatter-add over ALL samples (no live list; found_inf as in ngp_hash_bwd_f32_live): superseded by ngp_hash_bwd_f32_live
 * and the LDS-sliced form. */
int ngp_hash_bwd_f32_ex(const float* xyzs, const float* dout, const ngp_hash_levels* lv, int n_max,
                        const int32_t* n_dev, int normalize, float lo, float hi, int enc_pairs, float* dtable,
                        int32_t* found_inf, void* stream);

/* diagnostics: per-TASK task word + wall-clock stamps into a device buffer of 8 * 1536 uint64 (one 8-word row per task of the
 * plan, at most 1536 tasks; NULL = off, the default) */
int ngp_hash_bwd_sliced_debug(void* device_buffer);

/* host-side introspection of the task plan (no GPU): tasks[k] = level | slice << 4 | replica << 10; XCD x owns
 * tasks[xoff[x] .. xoff[x] + xlen[x]); returns the number of tasks or -2 when the level table cannot be expressed */
int ngp_hash_bwd_sliced_plan(const ngp_hash_levels* lv, uint16_t* tasks, int max_tasks, uint16_t* xoff /*[8]*/, uint16_t* xlen /*[8]*/,
                             uint8_t* nrep /*[NGP_MAX_LEVELS]*/, uint32_t* merge_mask, uint32_t* single_mask);

/* The count pass of the march WITHOUT the coarse-occupancy shortcut and with a mandatory hits_t (round 1's form): ngp_march_train_count_ex with
 * coarse = NULL is the same launch. */
int ngp_march_train_count(const float* rays_o, const float* rays_d, const float* hits_t,
                          const uint8_t* density_bitfield, const float* noise,
                          int cascades, int grid_size, float scale, float exp_step_factor,
                          int max_samples, int n_rays,
                          float* stage /*[n*max_samples,2] (t,dt)*/, int32_t* counts /*[n]*/,
                          void* stream);

/* The fused MLP backward over ALL samples [0, *n_dev) (no live list): superseded by ngp_mlp_bwd_live / ngp_mlp_bwd_live_parts. */
int ngp_mlp_bwd_ex(const float* enc, const float* dirs, const uint16_t* wpack, const float* dsigmas,
                   const uint16_t* drgbs, int n_max, const int32_t* n_dev, int enc_pairs, float* d_enc, float* dW,
                   int32_t* found_inf, void* stream);

/* ngp_hash_bwd_sliced_main_adam_step (include/ngp_hip.h) with the step's scalar bookkeeping done by the caller (ngp_train_prologue in front of the
 * launch): round 5's first form, kept for the A/B in tests/test_gpu_flush_adam.py; FusedTrainer launches the _step form. */
int ngp_hash_bwd_sliced_main_adam(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                  float* dtable, const void* workspace, long long workspace_bytes, const float* mlp_dw_parts,
                                  int n_parts, float* mlp_dw, float* table, float* table_m, float* table_v, uint16_t* table_bf16,
                                  const float* state_f, const int32_t* state_i, float beta1, float beta2, float eps,
                                  void* stream);

/* ngp_hash_bwd_sliced_main restricted to the levels in `level_mask` (bit l = level l), optionally on at most max_blocks persistent
 * workgroups (0 = as many as CUs): a caller that exchanges the gradient between ranks launches the fine levels first and sends
 * their part of dtable while the coarse levels are still being accumulated.  Launches over disjoint masks add up to the full
 * scatter-add; they share the prepass's workspace and must run one after the other on one stream. */
int ngp_hash_bwd_sliced_main_levels(const float* dout, const ngp_hash_levels* lv, int n_max, const int32_t* n_dev, int enc_pairs,
                                    float* dtable, int32_t* found_inf, const void* workspace, long long workspace_bytes,
                                    unsigned int level_mask, int max_blocks, void* stream);

/* ---- direct gradient exchange over peer memory (csrc/exchange.hip; round 6 prototype, SURVEY section 8e: "use the direct (all-links) reduce-scatter
 * + all-gather form rather than ring").  peer_dst / peer_flags are HOST arrays of `world` device pointers into the peers' allocations (this rank's own
 * entry included), mapped with hipIpc by the caller (ngp_hip/p2p.py).  Functional on one device only so far -- see the file header.
 * ngp_p2p_push: slice k of src (n_per_peer elements of elem_bytes = 4 or 2; broadcast != 0: the same n_per_peer elements for everybody) goes to
 *   peer_dst[k] + dst_offset, then peer_flags[k][rank] = step with system-scope release.  done: one uint32 of local memory, 0 between launches.
 * ngp_p2p_wait: bounded wait (max_spins polls per block; on expiry err[0] = 1 and nothing is written) for flags[0 .. world) >= step; reduce != 0:
 *   out[0 .. n) = scale * (sum of the `world` rows of inbox [world, n] f32). */
int ngp_p2p_max_peers(void);
int ngp_p2p_push(const void* src, long long n_per_peer, int elem_bytes, int world, void* const* peer_dst, int32_t* const* peer_flags,
                 long long dst_offset, int broadcast, int rank, int step, uint32_t* done, void* stream);
int ngp_p2p_wait(const int32_t* flags, int world, int step, long long max_spins, const float* inbox, long long n, float scale, int reduce,
                 float* out, int32_t* err, void* stream);

#ifdef __cplusplus
}
#endif
#endif
