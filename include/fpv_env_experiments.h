/*
 * fpv_env_experiments.h — entry points of libfpv_env_experiments.so (the library built with -DFPV_EXPERIMENTS=1) that the product
 * library does not export.  They are measured alternatives that lost, kept buildable and tested, out of the way of
 * the product interface:
 *   - fpv_rollout_steps: the whole rollout as one persistent launch (grid-wide barrier + statistics merge between steps);
 *   - fpv_policy_set_gemm(.., 2): the 16-wave policy kernel on 16x16x4 tiles (fpv_policy_w16.hip);
 *   - fpv_policy_io.narrow: 256-thread policy launches (the critic beside the env step);
 *   - the in-launch statistics merge of fpv_step_fused (fpv_step_extras.obs_rms / ret_rms folded by the launch's last workgroup;
 *     the product library serves the same request with the merge kernel behind the step).
 * Both libraries export everything include/fpv_env.h declares.
 */
#ifndef FPV_ENV_EXPERIMENTS_H
#define FPV_ENV_EXPERIMENTS_H

#include "fpv_env.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a whole rollout of T steps in ONE persistent launch ----
 * fpv_rollout_step T times without leaving the GPU: one 512-thread workgroup per 256 drones keeps the policy's weight image
 * in LDS and its drones' state in registers across the steps; between steps the workgroups meet at a grid-wide barrier
 * behind which each folds the step's moment sums into VecNormalize's statistics (what SB3's VecNormalize.step_wait does
 * before the next observation is normalised) — fpv_rollstep.hip.  Results are bit-identical to T calls of fpv_rollout_step
 * (and so to fpv_policy_forward + fpv_step_fused + the statistics merge, step by step).
 * Buffers are time-major, step t at base + t * num_envs * width.  On return: obs holds obs_T, terminated / truncated the
 * last step's flags, partials[(T - 1) & 1] the LAST step's moment sums (num_envs / 256 rows), still to be folded by the
 * caller (fpv_moments_merge_to with the statistics in rms_out[(T - 1) & 1] as input when T > 1, rms when T == 1).
 * Needs what fpv_rollout_step needs, and num_envs / 256 workgroups resident at once (at most one per CU; the launch is a
 * cooperative one and fails with FPV_ERR_HIP otherwise).  Every wait inside is bounded: if one times out (status word set
 * to 1) the rollout is invalid and the handle's state undefined. */
typedef struct {
    int32_t T;               /* steps */
    int32_t reserved;
    float* obs;              /* [n][20] raw observations: obs_0 in, obs_T out */
    float* obs_n;            /* [T][n][20] normalised observations (may be NULL) */
    float* action;           /* [T][n][4] sampled actions (may be NULL) */
    float* value;            /* [T][n] */
    float* logp;             /* [T][n] (may be NULL) */
    float* reward;           /* [T][n] raw rewards */
    uint8_t* done;           /* [T][n] terminated | truncated of step t (may be NULL) */
    uint8_t* terminated;     /* [n] */
    uint8_t* truncated;      /* [n] */
    float* ret;              /* [n] discounted-return accumulators (may be NULL) */
    float gamma, clip_obs, eps_norm;
    float reserved1;
    uint64_t seed, id_offset;        /* the policy's sampling stream, as fpv_policy_io */
    const double* rms;       /* [41] observation statistics entering step 0, or NULL: observations used as they are */
    const double* ret_rms;   /* [3] return statistics entering step 0, or NULL */
    double* rms_out[2];      /* the statistics after folding step t's sums go to rms_out[(t + 1) & 1], t = 0 .. T - 2 */
    double* ret_rms_out[2];
    double* snapshots;       /* [T][snapshot_stride]: row t as fpv_moments_merge's snapshot after folding step t (t <= T - 2); may be NULL */
    int64_t snapshot_stride; /* in doubles, >= FPV_PARTIAL_WORDS */
    double* partials[2];     /* two tables of num_envs / 256 rows of FPV_PARTIAL_WORDS doubles */
    uint32_t* status;        /* device word, written only on a timeout (zero it beforehand) */
} fpv_rollout_io;
int fpv_rollout_steps(fpv_env* env, fpv_policy* pol, const fpv_rollout_io* io, void* stream);

#ifdef __cplusplus
}
#endif
#endif
