/*
 * fpv_env.h — C ABI of the MI355X-native vectorised quadrotor hover environment and the
 * PPO rollout primitives (GAE, running-moment normaliser).
 *
 * The reference has no FFI: its replaceable seam is two duck-typed Python protocols
 * (SURVEY.md §8b).  Each entry point below names the reference interface it stands
 * under; the Python facades in fpv_drone_rl_agent_amd/ present those protocols and call
 * only these symbols (ctypes).  Plain pointers and sizes, no torch / HIP types: a
 * `stream` argument is a hipStream_t passed as void* (NULL = the default stream).
 *
 * All array arguments are DEVICE pointers unless a name ends in `_host`.  Calls are
 * stream-ordered and asynchronous with respect to the host; no entry point except
 * fpv_create / fpv_destroy / fpv_set_tick / fpv_get_tick / *_host synchronises.
 * One fpv_env per (process, device); a handle is not re-entrant.
 *
 * Return value: 0 on success, a negative fpv_status otherwise; fpv_last_error() then
 * holds a thread-local message.  HIP errors are captured, never abort().
 */
#ifndef FPV_ENV_H
#define FPV_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout version of the structs and enums below.  fpv_version() returns the value the library was built with: a host
 * compares the two before its first call (fpv_step_extras grew and fpv_detector was renumbered between 1 and 2, so a host
 * built against another header would pass structs of another size).  History: 1 = rounds 1-2; 2 = round 3
 * (FPV_DETECTOR_PIXEL = 0 is the default, debug hooks moved to fpv_env_debug.h); 3 = fpv_ledger_scan(status_out),
 * fpv_policy_io.scatter_*, the whole-rollout launch moved to fpv_env_experiments.h; 4 = round 4: fpv_ppo_io.reserved1 became
 * fpv_ppo_io.path (same layout; 0 now selects the one-kernel gradient), fpv_ppo_workspace_bytes covers every path; 5 = round 5:
 * fpv_ledger_pack and fpv_ledger_scan_finalize added (no layout changed); 6 = fpv_shuffle_indices added (no layout changed);
 * 7 = fpv_adam_clip_step_gated and fpv_ppo_kl_gate added, fpv_rollout_step serves fpv_policy_io.nets = 1 (no layout changed). */
#define FPV_ABI_VERSION 7

#define FPV_STATE_WORDS 36 /* docs/SPEC.md §2 canonical order */
#define FPV_OBS_WORDS 20   /* simulation/hover.py:63-70 */
#define FPV_ACT_WORDS 4    /* simulation/hover.py:59-61 */

typedef enum {
    FPV_OK = 0,
    FPV_ERR_ARG = -1,
    FPV_ERR_HIP = -2,
    FPV_ERR_ALLOC = -3,
    FPV_ERR_UNSUPPORTED = -4
} fpv_status;

typedef enum { FPV_F32 = 0, FPV_F64 = 1 } fpv_precision;

/* FPV_TASK_HOVER: simulation/hover.py (state 36, obs 20, action 4).  FPV_TASK_YAW: the yaw-setpoint task of
 * simulation/yaw.py as completed by docs/SPEC.md §10 (state 44, obs 12, action 1, optional domain randomisation). */
typedef enum { FPV_TASK_HOVER = 0, FPV_TASK_YAW = 1 } fpv_task;
#define FPV_YAW_STATE_WORDS 44
#define FPV_YAW_OBS_WORDS 12   /* simulation/yaw.py:39-45 */
#define FPV_YAW_ACT_WORDS 1    /* simulation/yaw.py:37 */

/* info bits of a done-ledger entry; ep_len is stored in bits 8..31 */
#define FPV_INFO_TERMINATED 1u    /* hover.py:281,290 */
#define FPV_INFO_TRUNCATED 2u     /* hover.py:275-276 */
#define FPV_INFO_OUT_OF_BOUNDS 4u /* hover.py:280 */
#define FPV_INFO_ON_FLOOR 8u      /* hover.py:289 */

typedef struct fpv_env fpv_env;

/* Constructor arguments.  Stands under QuadXHoverEnv.__init__(flight_mode=0, agent_hz=40,
 * render=False) (simulation/hover.py:11-70) for num_envs drones at once, i.e. under
 * make_vec_env(make_env, n_envs=...) (simulation/train_hover.py:36-42). */
typedef struct {
    int64_t num_envs;        /* drones owned by this handle (this rank's shard) */
    int32_t device;          /* HIP device ordinal */
    int32_t precision;       /* fpv_precision: arithmetic + I/O type of this handle */
    int32_t floor_check;     /* 1 = render=False (hover.py:283); 0 disables the floor rule */
    int32_t task;            /* fpv_task */
    double noise_ratio;      /* cf2x.yaml:5 motor noise (0.02); 0 = noise-free */
    uint64_t seed;           /* Philox key (docs/SPEC.md §9) */
    uint64_t id_offset;      /* global id of drone 0: rank * num_envs */
    int64_t ledger_capacity; /* done-ledger rows (rounded up to a multiple of the tile count); 0 = 2 * num_envs */
    int32_t domain_rand;     /* FPV_TASK_YAW: per-drone mass / drag resampled at every reset (BASELINE cfg 5) */
    int32_t detector;        /* fpv_detector: how obs[7..15] of the hover task come out of the camera pose (hover.py:157-222) */
} fpv_config;

/* The reference renders a 128 x 128 image and runs OpenCV's contour calls on it (hover.py:157-222).  Neither a renderer
 * nor OpenCV is part of this path.  FPV_DETECTOR_PIXEL (the default: a zero-initialised config selects it) applies the
 * reference's rules on the pixel grid — coverage by pixel centres, border rule, integer boundingRect, contourArea of the
 * pixel contour, the four approxPolyDP vertices (docs/SPEC.md §6b); from the reset pose it reproduces hover.py:50-51's
 * target_area = 0.013 and target_ratio = 1.53.  FPV_DETECTOR_ANALYTIC projects the panel's corners and takes centre /
 * area / bounding-box ratio of the continuous quadrilateral (SPEC §6): cheaper, but per-cent level differences in
 * obs[7,8,11,14] and the reward's target term.  The yaw task's target_center is analytic either way (its sphere
 * detector is missing from the reference). */
typedef enum { FPV_DETECTOR_PIXEL = 0, FPV_DETECTOR_ANALYTIC = 1 } fpv_detector;

/* Device-side view of the done ledger (SB3 VecEnv `infos[i]["terminal_observation"]`,
 * `["TimeLimit.truncated"]`, Monitor `["episode"] = {r, l}` — SURVEY.md §8a A11).
 * The ledger is partitioned per 64-drone tile (= one wavefront of fpv_step): tile t owns rows
 * [t * rows_per_tile, (t + 1) * rows_per_tile) and its own counter count[t], so a wave that sees
 * episode ends compacts them with one ballot and no global atomic.  Rows of a tile are appended in
 * lane order; count[t] keeps counting past rows_per_tile (overflow rows are dropped). */
typedef struct {
    int64_t capacity;      /* tiles * rows_per_tile */
    int64_t rows_per_tile;
    uint32_t* count;       /* [tiles] rows appended per tile since the last fpv_ledger_clear */
    int32_t* idx;          /* [capacity] local drone index */
    uint32_t* info;        /* [capacity] FPV_INFO_* | ep_len << 8 */
    uint32_t* tag;         /* [capacity] low 32 bits of the agent tick of the step that ended the episode */
    void* ep_return;       /* [capacity] float or double by precision */
    void* obs;             /* [capacity][FPV_OBS_WORDS] terminal observation */
} fpv_ledger;

/* ---- lifecycle ---- */
int fpv_create(const fpv_config* cfg, fpv_env** out);
/* QuadXHoverEnv.close() — hover.py:363-365 */
int fpv_destroy(fpv_env* env);

/* QuadXHoverEnv.reset() — hover.py:72-114; VecEnv.reset().  mask (uint8[num_envs]) selects
 * drones, NULL = all.  obs [num_envs][20] receives the reset observation of the selected
 * drones (others untouched); obs may be NULL.  prev_action is preserved (hover.py:101).
 * Deviation from SURVEY.md §8(b)'s sketch, on purpose: there is NO per-call seed argument.  The reference's reset(seed)
 * seeds gymnasium's RNG only and never reaches the simulator (hover.py:73); here the reset state is a
 * constant (noise-free warm-up, docs/SPEC.md §8) and every random draw of the path (motor noise, the yaw task's
 * heading / airframe) is a counter-based Philox stream keyed by fpv_config.seed, the global drone id and the agent
 * tick — reproducible across calls, ranks and shardings without a mutable RNG state to reseed. */
int fpv_reset(fpv_env* env, const uint8_t* mask, void* obs, void* stream);

/* QuadXHoverEnv.step() — hover.py:334-358 for every drone, with VecEnv auto-reset.
 * actions [num_envs][4] raw (unscaled, unclipped); obs [num_envs][20]; reward [num_envs];
 * terminated / truncated uint8 [num_envs] (FPV_TASK_YAW: actions [num_envs][1], obs [num_envs][12] —
 * simulation/yaw.py:102-149).  Float handles only.
 * Deviation from SURVEY.md §8(b)'s sketch, on purpose: there is NO dense terminal_obs [num_envs][20] output.  SB3's
 * infos[i]["terminal_observation"] exists only for the few drones whose episode ended in this step; a dense array would
 * add 80 B per drone and step to a 382 B path for values that are almost always unused.  They go to the done ledger
 * instead (fpv_ledger above): one compacted row per episode end — drone index, FPV_INFO_* bits, episode length,
 * Monitor's episode return and the terminal observation — appended by the step's own wave with a ballot and no atomic;
 * fpv_ledger_scan / fpv_ledger_scatter consume it on the device (timeout bootstrap), FpvVecEnv.step_wait on the host. */
int fpv_step(fpv_env* env, const float* actions, float* obs, float* reward, uint8_t* terminated,
             uint8_t* truncated, void* stream);
/* Same for an FPV_F64 handle (trajectory-level parity runs). */
int fpv_step_f64(fpv_env* env, const double* actions, double* obs, double* reward,
                 uint8_t* terminated, uint8_t* truncated, void* stream);

/* fpv_step plus SB3 VecNormalize's per-step bookkeeping in the same kernel (simulation/train_hover.py:43):
 * extras.done [num_envs] = terminated | truncated; extras.ret [num_envs] is VecNormalize's discounted-return
 * accumulator (ret = ret * gamma + reward, cleared where an episode ended); extras.partials
 * [ceil(num_envs/64)][FPV_PARTIAL_WORDS] (capacity) receives fp64 sums
 * {sum obs[20], sum obs^2[20], sum ret, sum ret^2, 0, 0} of the returned observations and of ret, one row per
 * 64-drone tile or per group of four tiles: fpv_partial_rows() says how many rows are written (ask again after
 * fpv_set_pair_threshold); they are folded into the running statistics by fpv_moments_merge(rows, ...) or by
 * the next fpv_policy_forward (merge_* fields).  Any of the three may be NULL.  fp32 handles; the yaw task's rows are
 * 2 x 12 + 4 = 28 words ({sum obs[12], sum obs^2[12], sum ret, sum ret^2, 0, 0}), the hover task's 44. */
#define FPV_PARTIAL_WORDS 44
typedef struct {
    uint8_t* done;
    float* ret;
    double* partials;
    float gamma;
    int32_t reserved;
    /* Optional: RunningMeanStd.update inside this call.  obs_rms = double[2 O + 1] {mean, var, count}, ret_rms = double[3],
     * updated IN PLACE with this step's sums (batch count = num_envs); snapshot (double[FPV_PARTIAL_WORDS], may be NULL)
     * receives {obs statistics from word 0, ret statistics at 41..43} as they stand after the update.  Needs partials.
     * With four tiles per workgroup (num_envs % 256 == 0 and above the wave-pair threshold) the workgroup that finishes
     * last folds the rows inside the same launch (write-through rows + a ticket; fixed summation order); other launch
     * shapes run the merge kernel behind the step.  Either way the statistics are final when the call's work is done
     * on the stream, which is what the next fpv_policy_forward reads. */
    double* obs_rms;
    double* ret_rms;
    double* snapshot;
} fpv_step_extras;
int fpv_step_fused(fpv_env* env, const float* actions, float* obs, float* reward, uint8_t* terminated,
                   uint8_t* truncated, const fpv_step_extras* extras, void* stream);

int fpv_partial_rows(fpv_env* env, int64_t* rows);

/* RunningMeanStd.update for both VecNormalize statistics from the per-tile sums of fpv_step_fused:
 * obs_rms = double[41] {mean[20], var[20], count}, ret_rms = double[3] {mean, var, count} (either may be
 * NULL); snapshot (optional double[FPV_PARTIAL_WORDS]) receives {obs_rms[41], ret_rms[3]} after the merge. */
int fpv_moments_merge(int64_t tiles, const double* partials, double batch_count, double* obs_rms,
                      double* ret_rms, double* snapshot, void* stream);
/* The same with separate inputs and outputs (in == out is allowed) and for any observation width O = obs_words <= 20
 * (12 for the yaw task): rows of 2 O + 4 words, obs statistics double[2 O + 1] = {mean[O], var[O], count}; the
 * snapshot holds the obs statistics from word 0 and the ret statistics at words 41..43. */
int fpv_moments_merge_to(int64_t tiles, int32_t obs_words, const double* partials, double batch_count,
                         const double* obs_rms_in, const double* ret_rms_in, double* obs_rms_out, double* ret_rms_out,
                         double* snapshot, void* stream);

/* VecNormalize.normalize_reward applied after the fact: out[t][i] = clip(raw[t][i] / sqrt(var_t + eps)) with
 * var_t = snapshots[t][42], the return variance as it stood after step t.  raw/out are [T][n]; out may alias raw. */
int fpv_reward_finalize(int64_t T, int64_t n, const float* raw, const double* snapshots, float clip, float eps,
                        float* out, void* stream);

/* Word counts of this handle's task: state blob row, observation row, action row. */
int fpv_dims(fpv_env* env, int32_t* state_words, int32_t* obs_words, int32_t* act_words);

/* Device address of the handle's agent-tick counter (uint64; keys in-kernel RNG streams, e.g. fpv_policy_io.tick). */
int fpv_tick_ptr(fpv_env* env, const uint64_t** out);

/* Canonical state blobs [num_envs][36] in the handle's precision (docs/SPEC.md §2); used by
 * parity tests and the synthetic random-init benchmark. */
int fpv_get_state(fpv_env* env, void* state_out, void* stream);
int fpv_set_state(fpv_env* env, const void* state_in, void* stream);

/* Agent-tick counter that keys the motor-noise RNG.  It lives on the device (one copy per
 * 64-drone tile, advanced by the step kernel itself) so that fpv_step is a single kernel and
 * can be captured into a HIP graph.  These two synchronise the stream. */
int fpv_set_tick(fpv_env* env, uint64_t tick, void* stream);
int fpv_get_tick(fpv_env* env, uint64_t* tick_host, void* stream);

int fpv_ledger_view(fpv_env* env, fpv_ledger* out);
int fpv_ledger_clear(fpv_env* env, void* stream);
/* The done ledger as ONE dense block, for a host that reads it after every step (FpvVecEnv's numpy mode under stock SB3:
 * infos[i]["terminal_observation"], ["episode"], hover.py:102's flags — simulation/train_hover.py:42-43): a header of 8 words —
 * out[0] = rows appended since the last fpv_ledger_clear (rows that no longer fitted a tile's partition included), out[1] = rows
 * packed here (<= max_rows), out[2] = words per row W = 4 + obs words, out[3] = rows held by the ledger — followed by the rows
 * {idx, info, tag, ep_return (float), obs[..]} in (tile, row) order, which for a ledger cleared before the step is the order of
 * the drone index.  obs_rms (float64 {mean[O], var[O], count}) != NULL: the terminal observation is written normalised,
 * clip((x - mean) / sqrt(var + eps_norm), +-clip_obs), the arithmetic of fpv_normalize — what SB3's VecNormalize.step_wait does
 * to infos[i]["terminal_observation"].  out holds 8 + max_rows * W words (device memory; the caller copies it to the host
 * together with the step's outputs).  One small launch; float handles. */
int fpv_ledger_pack(fpv_env* env, const double* obs_rms, float clip_obs, float eps_norm, uint32_t* out, int64_t max_rows, void* stream);
/* Start of a rollout in one launch: fpv_ledger_clear, the episode-start flags carried over from the end of the previous
 * rollout (starts_dst[i] = starts_src[i] for the handle's num_envs drones; both NULL to skip; 16-byte aligned), and — when
 * given — the words fpv_ledger_scan will accumulate into at the end of this rollout set to zero (scan_counters[4],
 * scan_sums[2]; both or neither; fpv_ledger_scan is then called with counters_are_zero = 1 and needs no launch of its own for
 * that).  A kernel, so that a captured rollout holds no memcpy / memset node. */
int fpv_rollout_begin(fpv_env* env, uint8_t* starts_dst, const uint8_t* starts_src, int32_t* scan_counters, double* scan_sums,
                      void* stream);
/* Replace the handle-owned ledger by caller-owned device buffers (same layout as fpv_ledger;
 * ep_return / obs in the handle's precision; count has ceil(num_envs / 64) entries and capacity
 * must equal that tile count times rows_per_tile).  The caller keeps them alive until
 * fpv_destroy or the next bind.  Lets a host runtime view the ledger without copies. */
int fpv_ledger_bind(fpv_env* env, const fpv_ledger* buffers);

/* ---- PPO rollout primitives ---- */

/* SB3 RolloutBuffer.compute_returns_and_advantage: rewards / values / episode_starts are
 * [T][n] (time-major), last_values / last_dones [n]; writes advantages / returns [T][n].
 * stats (optional, device double[3]) receives {count, sum(A), sum(A^2)} of this shard —
 * the payload of the one all-gather per update.  mean_std (optional, device double[2]; needs stats)
 * receives {mean, population std} of THIS shard's advantages in the same launch: what fpv_advantage_stats derives from the
 * gathered payload when there is one rank only.
 * The handle-less primitives (fpv_gae, fpv_normalize*, fpv_moments_merge*) share one reduction scratch per device: calls on
 * different streams of one device must not overlap. */
int fpv_gae(int64_t T, int64_t n, const float* rewards, const float* values,
            const uint8_t* episode_starts, const float* last_values, const uint8_t* last_dones,
            float gamma, float lam, float* advantages, float* returns, double* stats, double* mean_std,
            void* stream);

/* SB3 VecNormalize running moments (train_hover.py:43): merge the batch x [n][d] into
 * rms = double[2*d + 1] = {mean[d], var[d], count} (parallel-variance merge), when
 * update != 0, then write y = clip((x - mean) / sqrt(var + eps), +-clip) (y may alias x,
 * or be NULL to only update).  d <= 32. */
int fpv_normalize(int64_t n, int32_t d, const float* x, float* y, double* rms, int32_t update,
                  float clip, float eps, void* stream);

/* VecNormalize reward path: ret = ret * gamma + r; merge ret into rms = double[3]
 * {mean, var, count} when update != 0; r_out = clip(r / sqrt(var + eps), +-clip);
 * ret[done] = 0.  r_out may alias r. */
int fpv_normalize_reward(int64_t n, const float* r, const uint8_t* done, float* ret, float* r_out,
                         double* rms, int32_t update, float gamma, float clip, float eps,
                         void* stream);

/* The single collective of the rollout path is an all-gather of fpv_gae's stats = {count, sum A, sum A^2} per rank
 * (SURVEY 8e).  gathered [world][stride] (stride >= 3) -> mean_std[2] = {mean, population std} of the advantages over all
 * ranks, added in rank order (every rank derives the same bits). */
int fpv_advantage_stats(const double* gathered, int32_t world, int32_t stride, double* mean_std, void* stream);
/* VecNormalize's statistics ride in the same payload: gathered row r = {adv stats[3], obs_rms[2 O + 1], ret_rms[3]} as
 * rank r holds them after a rollout that every rank started from base_obs_rms / base_ret_rms.  What each rank added is recovered
 * from (base, its row) and merged into the base in rank order (RunningMeanStd.update_from_moments): obs_rms / ret_rms (may alias
 * the bases) receive the statistics of the union of all ranks' samples, the same bits on every rank.  Either statistics may be
 * left out (both pointers NULL).  simulation/train_hover.py:43's VecNormalize over ALL drones, once per update. */
int fpv_rank_merge(const double* gathered, int32_t world, int32_t stride, int32_t obs_words, const double* base_obs_rms,
                   const double* base_ret_rms, double* obs_rms, double* ret_rms, void* stream);

/* ---- timeout bootstrap (SB3 OnPolicyAlgorithm.collect_rollouts: reward += gamma * V(terminal observation) where an
 * episode was truncated, hover.py:275, and not terminated) without a host round trip ----
 * fpv_ledger_scan walks the handle's done ledger after a rollout of T agent steps: counters[4] = {K, valid rows,
 * episodes appended (including rows that did not fit), flags: 1 ledger overflow | 2 tag outside the rollout | 4 more
 * than cap list entries}, sums[2] = {sum of episode returns, sum of episode lengths} (Monitor statistics), and the K
 * truncated-not-terminated entries as a dense list: t_idx[k] (step within the rollout), d_idx[k] (drone), tobs[k][O]
 * (terminal observation, normalised with snapshots[t_idx[k]] = {mean[O], var[O], ...} when snapshots != NULL).
 * The value net then runs over tobs with fpv_policy_io.n_dev = &counters[0], and fpv_ledger_scatter adds
 * gamma * values[k] to rewards[t_idx[k]][d_idx[k]] ([T][n]).  The list order is arbitrary, the result is not.
 * counters_are_zero != 0: counters / sums were zeroed by this rollout's fpv_rollout_begin (else a small launch zeroes them here).
 * status_out (optional, 32 bytes, 8-byte aligned): receives {counters[4], sums[2]} as they stand at the end of the launch, written
 * by the workgroup that finishes last with system-scope stores — so it may be pinned HOST memory that the device can address
 * (hipHostMalloc / a pinned torch tensor): the host reads the episode statistics and the overflow flags after waiting for the
 * launch, with no device-to-host copy queued behind the rollout. */
int fpv_ledger_scan(fpv_env* env, const double* snapshots, int64_t snapshot_stride, int32_t T, float clip_obs, float eps_norm,
                    float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                    int32_t counters_are_zero, void* status_out, void* stream);
/* fpv_ledger_scan and fpv_reward_finalize(T, num_envs, raw, reward_snapshots, clip_reward, eps_reward, rewards) as ONE launch: the
 * scan's workgroups walk the ledger, further workgroups of the same grid scale the rollout's rewards (rewards[t][i] =
 * clip(raw[t][i] / sqrt(var_t + eps_reward)), var_t = reward_snapshots[t][42]; raw / rewards [T][num_envs], rewards may alias raw).
 * The two are independent of each other and of V(s_T); as two launches the scaling stood in front of the scan on the rollout's
 * tail (SB3 collect_rollouts' end).  Same bits as the two calls. */
int fpv_ledger_scan_finalize(fpv_env* env, const double* snapshots, int64_t snapshot_stride, int32_t T, float clip_obs, float eps_norm,
                             float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                             int32_t counters_are_zero, void* status_out, const float* raw, const double* reward_snapshots,
                             int64_t reward_snapshot_stride, float clip_reward, float eps_reward, float* rewards, void* stream);
int fpv_ledger_scatter(const int32_t* counters, const int32_t* t_idx, const int32_t* d_idx, const float* values, float gamma,
                       float* rewards, int64_t n, int64_t cap, void* stream);

/* ---- fused actor-critic forward (SB3 MlpPolicy, net_arch pi/vf [128,128], tanh) ----
 * Stands under model.policy.forward(obs) / model.predict(obs, deterministic) of the SB3 learner in
 * simulation/train_hover.py:47-60 and simulation/test_hover.py:17 during rollout collection; the
 * parameters stay torch tensors owned by the caller (the PPO update differentiates through torch). */
typedef struct fpv_policy fpv_policy;

int fpv_policy_create(int32_t device, fpv_policy** out);
int fpv_policy_destroy(fpv_policy* pol);
/* weights: HOST array of 13 DEVICE pointers to contiguous fp32 nn.Linear tensors, in this order:
 * pi.0.weight[128][20], pi.0.bias[128], pi.2.weight[128][128], pi.2.bias[128], action_net.weight[4][128],
 * action_net.bias[4], log_std[4], vf.0.weight, vf.0.bias, vf.2.weight, vf.2.bias, value_net.weight[1][128],
 * value_net.bias[1].  Repacks them into the kernel's MFMA-ordered image (call after every update). */
int fpv_policy_load(fpv_policy* pol, const float* const* weights, void* stream);
/* Observation / action widths (default 20 / 4, simulation/hover.py:59-70; 12 / 1 for simulation/yaw.py:37-45).  in_dim a
 * multiple of 4 up to 24, act_dim 1..4; first-layer weights are then [128][in_dim], the action head [act_dim][128], and
 * the obs / action row buffers of fpv_policy_io have those widths.  Call before fpv_policy_load. */
int fpv_policy_set_dims(fpv_policy* pol, int32_t in_dim, int32_t act_dim);

typedef struct {
    int64_t n;               /* rows */
    const float* obs;        /* [n][20] raw observations */
    const double* rms;       /* VecNormalize statistics {mean[20], var[20], count} or NULL (obs used as is) */
    const float* eps_in;     /* [n][4] N(0,1) draws, or NULL: Philox4x32-10 in-kernel */
    const uint64_t* tick;    /* device counter keying the in-kernel draws (NULL = 0) */
    float* obs_n;            /* [n][20] normalised observations out (may be NULL) */
    float* action;           /* [n][4] a = mu + sigma * eps (may be NULL) */
    float* clipped;          /* [n][4] clip(a, -1, 1): what fpv_step receives (may be NULL) */
    float* value;            /* [n] */
    float* logp;             /* [n] log pi(a) (may be NULL) */
    float* eps_out;          /* [n][4] the draws used (may be NULL) */
    uint64_t seed, id_offset;
    float clip_obs, eps_norm;
    int32_t deterministic;   /* 1: a = mu */
    int32_t nets;            /* 0 or 3: pi and V; 1: pi only (action / log-prob / clip outputs); 2: V only (value) */
    int32_t narrow;          /* 1: 256-thread workgroups (one wave per SIMD) so that a VALU-bound kernel such as
                                fpv_step can co-execute on another stream; 0: 512-thread workgroups */
    int32_t reserved;
    /* Optional in-launch VecNormalize update (all NULL / 0 to skip): fold the tile sums that the PREVIOUS fpv_step_fused
     * wrote (extras.partials, merge_rows <= 1024 rows) into the statistics before normalising — what fpv_moments_merge
     * does, without its launch on the rollout's critical path.  `rms` (and `ret_rms`) are read; the merged statistics
     * are written to rms_out / ret_rms_out, which must be different buffers; merge_snapshot as in fpv_moments_merge.
     * Rows are 2 in_dim + 4 words wide (fpv_policy_set_dims).  Needs narrow = 0. */
    const double* merge_partials;
    int64_t merge_rows;
    double merge_count;      /* drones behind the sums (batch count of the RunningMeanStd update) */
    double* rms_out;         /* [41] */
    const double* ret_rms;   /* [3] or NULL */
    double* ret_rms_out;     /* [3] */
    double* merge_snapshot;  /* [FPV_PARTIAL_WORDS] or NULL */
    const int32_t* n_dev;    /* optional DEVICE row count: rows = min(n, *n_dev) (lists whose length only the device knows) */
    /* Optional scatter of the values (V evaluated; NULL to skip): scatter_out[scatter_t[k] * scatter_stride + scatter_d[k]] +=
     * scatter_scale * value[k] for every evaluated row k — fpv_ledger_scatter in the launch that produces the values (the timeout
     * bootstrap: t_idx / d_idx of fpv_ledger_scan, the [T][n] reward table, gamma).  Every (t, d) pair must be distinct. */
    const int32_t* scatter_t;
    const int32_t* scatter_d;
    float* scatter_out;
    int64_t scatter_stride;
    float scatter_scale;
    int32_t reserved2;
} fpv_policy_io;

int fpv_policy_forward(fpv_policy* pol, const fpv_policy_io* io, void* stream);
/* Hidden-layer GEMM arithmetic: 0 (default) = v_mfma_f32_32x32x2_f32, bit-exact fp32 fma chains; 1 = each fp32
 * operand cut into three bf16 pieces and the six piece products of order <= 4 summed on v_mfma_f32_32x32x16_bf16
 * in fp32 accumulators (error at the level of one fp32 rounding, 2.7x fewer matrix-pipe cycles; built for the two tasks'
 * widths, 20 -> 4 and 12 -> 1 — a handle of other widths keeps answering with mode 0's kernel); 2 = the arithmetic of
 * mode 0 on v_mfma_f32_16x16x4_f32 with four waves per SIMD (fpv_policy_w16.hip) for the rollout's launch shape (both
 * nets, in-kernel sampling, no merge prologue): another summation order, so it agrees with mode 0 to rounding, not bit
 * for bit.  Kept as the measured alternative: it is not faster — vector instructions do not co-execute
 * with fp32-input MFMA on gfx950 (tools/mfma_mix), so the tanh work bounds both shapes alike. */
int fpv_policy_set_gemm(fpv_policy* pol, int32_t mode);

/* ---- one rollout step in one launch: fpv_policy_forward + fpv_step_fused of the same drones ----
 * What SB3's collect_rollouts does per step over VecNormalize(VecEnv) (simulation/train_hover.py:42-60): policy forward on
 * obs_t, Box clip, env.step with auto-reset, VecNormalize bookkeeping — by ONE kernel whose 512-thread workgroups evaluate
 * pi / V for 256 drones and then step those drones (fpv_rollstep.hip): the action never leaves the chip and the env step
 * shares the launch.  `io` as for fpv_policy_forward with io->obs = the handle's raw observation rows obs_t, which are
 * REPLACED by obs_{t+1} (io->n = num_envs; eps_in, eps_out, deterministic, narrow, n_dev, V alone are not served; nets = 1 —
 * pi alone, io->value unused: the critic deferred to one fpv_policy_forward over the stored obs_n rows, same bits — in gemm mode 1); reward / terminated / truncated / extras as for fpv_step_fused (extras->obs_rms / ret_rms / snapshot must be
 * NULL: the statistics update rides in the NEXT call's io->merge_* fields).  One row of moment sums per workgroup:
 * extras->partials receives num_envs / 256 rows, and because every workgroup of call t + 1 reads call t's rows while it
 * writes its own, the caller alternates between two tables (io->merge_partials != extras->partials).
 * Results are bit-identical to fpv_policy_forward followed by fpv_step_fused.  Needs: an fp32 handle (hover or yaw task) with
 * noise_ratio > 0, num_envs a multiple of 256, the policy widths of the task (20 / 4, 12 / 1: fpv_policy_set_dims) in gemm
 * mode 0 or 1 (rollout_step_kernel / rollout_step_split_kernel); FPV_ERR_UNSUPPORTED otherwise.  Rows of moment sums are 2 x obs_words + 4 doubles wide, as for fpv_step_fused. */
int fpv_rollout_step(fpv_env* env, fpv_policy* pol, const fpv_policy_io* io, float* reward, uint8_t* terminated,
                     uint8_t* truncated, const fpv_step_extras* extras, void* stream);

/* The whole-rollout persistent launch (fpv_rollout_steps) is declared in fpv_env_experiments.h: measured slower than one launch
 * per step, it is built into libfpv_env_experiments.so only. */


/* ---- PPO minibatch gradient (SB3 PPO.train: clipped surrogate + vf_coef * MSE value loss - ent_coef * entropy) ----
 * The learner north_star puts in place of the SAC of simulation/train_hover.py:47-60.  One call = forward of both nets
 * over `rows` transitions, the loss derivatives, and the backward pass, as hand-written MFMA kernels (fpv_ppo.hip);
 * the gradient of the MEAN loss over the rows is WRITTEN (not accumulated) to grads[]; optimiser, clipping and the
 * cross-rank all-reduce stay with the caller.  weights / grads: 13 DEVICE pointers each, in fpv_policy_load's order.
 * Deterministic (no atomics).  Everything is stream-ordered; nothing is allocated (graph-capturable). */
#define FPV_PPO_ONE_KERNEL 0
#define FPV_PPO_TWO_KERNELS 1
#define FPV_PPO_TWO_KERNELS_BF16X6 2
#define FPV_PPO_TWO_KERNELS_BF16X3 3
typedef struct {
    int64_t rows;             /* minibatch size M */
    const int64_t* index;     /* [M] device row indices into the arrays below, or NULL: rows 0..M-1 */
    const float* obs;         /* [*][in_dim] normalised observations (rollout buffer) */
    const float* actions;     /* [*][act_dim] unclipped sampled actions */
    const float* old_logp;    /* [*] */
    const float* advantages;  /* [*] as the loss should see them (normalise beforehand) */
    const float* returns;     /* [*] */
    float clip_range, vf_coef, ent_coef, reserved0;
    const float* weights[13];
    float* grads[13];
    float* stats;             /* [4] out: policy-gradient loss, value loss (MSE), approx KL, clip fraction */
    void* workspace;          /* device scratch, >= fpv_ppo_workspace_bytes(rows), 16-byte aligned (round 5: 384 KB more than   */
                              /* before — the one-kernel path's image of W2 cut into bf16 pieces; ask again, do not cache a size) */
    int64_t workspace_bytes;
    int32_t storage;          /* what the two-kernel paths write between their kernels: 0 = fp32 words (default), 1 = bf16 —  */
                              /* half the HBM traffic of the update, gradients within ~1e-3 of their max-norm.  (A storage */
                              /* format, not an arithmetic: which pipe multiplies is `path`'s business.)                   */
    int32_t path;             /* FPV_PPO_ONE_KERNEL (0, default): forward, backward and the weight gradients of a row in  */
                              /* one kernel, the activations / deltas never leave the CU (storage must be 0).  Its three   */
                              /* 128 x 128 products per row run as 6-term bf16 splits of the fp32 operands with fp32      */
                              /* accumulation and its tanh is the fast form: fp32-LEVEL (golden error 4.5e-7 of a tensor's  */
                              /* max-norm), not the exact fp32-input MFMA of path 1.                                       */
                              /* FPV_PPO_TWO_KERNELS (1): they pass through memory between two kernels (`storage`).        */
                              /* FPV_PPO_TWO_KERNELS_BF16X6 (2): as 1 with the weight-gradient products on the bf16 MFMA, */
                              /* every fp32 operand cut into three bf16 pieces, six piece products per product (fp32      */
                              /* accumulation, fp32-level error; storage 0 only).  Same results to fp32 summation order.  */
                              /* FPV_PPO_TWO_KERNELS_BF16X3 (3): as 2 with three piece products — NOT fp32-level (weight  */
                              /* gradients to ~5e-6 of their max-norm); kept so that its error is measured, not guessed.   */
} fpv_ppo_io;

int fpv_ppo_workspace_bytes(fpv_policy* pol, int64_t rows, int64_t* bytes);
int fpv_ppo_grad(fpv_policy* pol, const fpv_ppo_io* io, void* stream);

/* torch.nn.utils.clip_grad_norm_(max_grad_norm) + torch.optim.Adam.step() (no weight decay, no amsgrad) on ONE flat
 * fp32 tensor of n elements, as a single launch (SB3 PPO.train's optimiser step, max_grad_norm 0.5, Adam eps 1e-5).
 * step: device int64[2], zero-initialised by the caller: {step count (incremented), launch-internal counter}; grad_norm: optional device float receiving the norm before clipping;
 * max_grad_norm <= 0 disables clipping.  grad is not modified. */
int fpv_adam_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                       float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                       void* stream);
/* SB3 PPO.train's target_kl rule ("stop the update when approx_kl > 1.5 target_kl") without a host round trip per epoch: every epoch of
 * an update is queued at once; `gate` (device int32[2], zeroed by the caller before the update: {stopped, epochs that counted}) is read
 * by fpv_adam_clip_step_gated — a launch that finds gate[0] != 0 changes nothing: parameters, moments and step count stay as the last
 * counted epoch left them — and written by fpv_ppo_kl_gate behind an epoch's minibatches: an epoch that started unstopped counts
 * (gate[1] += 1), and sets gate[0] when the mean of its n_minibatches approx_kl values (epoch_stats[k][2], the statistics rows of
 * fpv_ppo_io.stats) is above `limit` (= 1.5 target_kl).  The host reads gate[1] with the statistics, whenever it wants them. */
int fpv_adam_clip_step_gated(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                             float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                             const int32_t* gate, void* stream);
int fpv_ppo_kl_gate(const float* epoch_stats, int32_t n_minibatches, float limit, int32_t* gate, void* stream);

/* The row order of a PPO epoch (SB3 PPO.train draws np.random.permutation(buffer rows) per epoch — stable_baselines3's
 * RolloutBuffer.get, the learner SURVEY §8(f) N1 puts in place of train_hover.py:47-60's): out[i], i = 0..n-1, is a permutation of
 * 0..n-1 chosen by (seed, counter) — a keyed bijection (8-round Feistel network on the index bits with cycle walking), one thread per
 * element, instead of a sort of random keys.  Every call with another counter gives another order; the same (seed, counter) the same. */
int fpv_shuffle_indices(int64_t* out, int64_t n, uint64_t seed, uint64_t counter, void* stream);

/* Diagnostic hooks (phase stamps, launch-shape switches, calibration launches) are declared in fpv_env_debug.h: they are not
 * part of the product interface. */

const char* fpv_last_error(void);
int fpv_version(void);

#ifdef __cplusplus
}
#endif
#endif
