/*
 * fpv_oracle.h — CPU restatement of the hover env-step path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker for the HIP path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (fpv_drone_rl_agent_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * PARITY UNPINNED versus PyFlyt 0.21.0 / pybullet 3.2.7: the reference delegates the
 * rigid-body arithmetic to those packages (simulation/uv.lock:807-814), neither is on
 * disk, and the reference ships no tests or golden vectors.  What
 * IS pinned: every rule simulation/hover.py states itself (action scaling :337-341,
 * angle wrap :229, Euler->quat :233, obs layout :253-267, reward/termination order
 * :274-332, episode bookkeeping :354-357) via the hand-derived known answers of
 * SURVEY.md §8c, checked in tests/test_oracle_kat.py.  The physics inside
 * Aviary.step() follows docs/SPEC.md, a build-defined written spec.
 *
 * Plain C99, double precision, one drone at a time, array-of-structs state.
 */
#ifndef FPV_ORACLE_H
#define FPV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPV_ORACLE_STATE_WORDS 36
#define FPV_ORACLE_OBS_WORDS 20
#define FPV_ORACLE_YAW_STATE 44 /* SPEC §10: 36 hover words + act_hist[4] + mass, Cd, k_pqr, spare */
#define FPV_ORACLE_YAW_OBS 12   /* yaw.py:39-45 */

/* canonical word order of docs/SPEC.md §2 */
typedef struct {
    double pos[3];
    double quat[4];  /* x y z w, body->world */
    double vel[3];   /* world */
    double omega[3]; /* body */
    double thr[4];
    double pid_int[3];
    double pid_prev[3];
    double prev_euler[3];
    double prev_center[2];
    double prev_area;
    double prev_ratio;
    double prev_action[4];
    double step_count;
    double ep_return;
} fpv_oracle_state;

typedef struct {
    int32_t floor_check;   /* hover.py:283 — 0 when render=True */
    int32_t domain_rand;   /* SPEC §10: resample mass / Cd / k_pqr at every yaw-task reset */
    double noise_ratio;    /* cf2x.yaml:5 (0.02) or 0 for noise-free KATs */
    uint64_t seed;         /* Philox key */
    uint64_t id_offset;    /* global id of drone 0 (rank offset) */
    int32_t detector;      /* 0: analytic projection (SPEC §6), 1: 128 x 128 pixel rules of hover.py:157-222 (SPEC §6b) */
    int32_t reserved;
} fpv_oracle_cfg;

/* info bits written to the done ledger and by fpv_oracle_reward */
#define FPV_ORACLE_TERM 1u
#define FPV_ORACLE_TRUNC 2u
#define FPV_ORACLE_OOB 4u
#define FPV_ORACLE_FLOOR 8u

/* ---- small pinned pieces (KAT targets) ---- */
void fpv_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void fpv_oracle_normal4(uint64_t seed, uint64_t drone_id, uint64_t tick, double g[4]);
double fpv_oracle_wrap(double d);                                   /* hover.py:229 */
void fpv_oracle_quat_from_euler(const double e[3], double q[4]);    /* hover.py:233 */
void fpv_oracle_euler_from_quat(const double q[4], double e[3]);
void fpv_oracle_scale_action(const double a[4], double sp[4]);      /* hover.py:337-341 */
/* SPEC §6: returns visible (0/1); center[2], area, ratio */
int fpv_oracle_camera(const double pos[3], const double quat[4], double center[2], double* area,
                      double* ratio);
/* SPEC §6b: the same through the 128 x 128 pixel rules of hover.py:157-222 (cv2's rules restated: parity unpinned) */
int fpv_oracle_camera_pixel(const double pos[3], const double quat[4], double center[2], double* area,
                            double* ratio);
/* test hook: dbg = row spans [128][2] (-1 = empty), [256..263] the 4 vertices (x, y), [264] arcLength, [265] contourArea [px^2],
 * [266] len(approx), [267] len(contours), [268] points of the chosen contour, [272 ...] those points (x, y), at most 512 */
#define FPV_ORACLE_PIXEL_DBG (272 + 1024)
int fpv_oracle_camera_pixel_debug(const double pos[3], const double quat[4], double center[2], double* area,
                                  double* ratio, double dbg[FPV_ORACLE_PIXEL_DBG]);
/* hover.py:180-213 from a 128 x 128 mask (0 / 1 bytes) on: border test, findContours, max contourArea, approxPolyDP, features */
int fpv_oracle_detect_mask(const uint8_t* mask, double center[2], double* area, double* ratio, double* dbg);
/* SPEC §7 / hover.py:274-332.  Returns reward, writes info bits. */
double fpv_oracle_reward(const double pos[3], const double euler[3], double omega_z,
                         const double center[2], double area, double ratio, int visible,
                         const double action[4], const double prev_action[4], double step_count,
                         int floor_check, uint32_t* info);

/* ---- one drone ---- */
void fpv_oracle_reset_state(fpv_oracle_state* s, double obs[FPV_ORACLE_OBS_WORDS]);     /* detector 0 */
void fpv_oracle_reset_state_d(fpv_oracle_state* s, double obs[FPV_ORACLE_OBS_WORDS], int detector);
/* one physics sub-step k (SPEC §4); pwm[4] is carried between the even and odd sub-step */
void fpv_oracle_substep(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double sp[4],
                        double pwm[4], int k, uint64_t drone_id, uint64_t tick);
/* one agent step WITHOUT auto-reset; returns info bits */
uint32_t fpv_oracle_step_one(const fpv_oracle_cfg* cfg, fpv_oracle_state* s, const double action[4],
                             uint64_t drone_id, uint64_t agent_tick, double obs[FPV_ORACLE_OBS_WORDS],
                             double* reward);

/* ---- batch (VecEnv semantics, SPEC §8) ----
 * state [n][36], actions [n][4], obs [n][20], reward [n], terminated/truncated [n].
 * Done ledger (ascending drone index): led_idx[], led_info[] (bits | ep_len << 8),
 * led_ret[], led_obs[][20]; *led_count is read as the current fill and advanced.
 * nthreads > 1 uses OpenMP over drones. */
void fpv_oracle_reset_batch(int64_t n, double* state, double* obs, const uint8_t* mask_or_null);   /* detector 0 */
void fpv_oracle_reset_batch_d(int64_t n, double* state, double* obs, const uint8_t* mask_or_null, int detector);
void fpv_oracle_step_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, const double* actions,
                           uint64_t agent_tick, double* obs, double* reward, uint8_t* terminated,
                           uint8_t* truncated, int64_t* led_count, int32_t* led_idx,
                           uint32_t* led_info, double* led_ret, double* led_obs, int nthreads);

/* ---- GAE (SB3 RolloutBuffer.compute_returns_and_advantage semantics) ----
 * rewards/values/episode_starts [T][n]; last_values/last_dones [n]; adv/returns [T][n]. */
void fpv_oracle_gae(int64_t T, int64_t n, const double* rewards, const double* values,
                    const uint8_t* episode_starts, const double* last_values,
                    const uint8_t* last_dones, double gamma, double lam, double* adv,
                    double* returns);

/* ---- yaw task (simulation/yaw.py skeleton + docs/SPEC.md §10); state [n][44], obs [n][12], actions [n] ---- */
void fpv_oracle_yaw_reset_state(const fpv_oracle_cfg* cfg, double* y, uint64_t drone_id, uint64_t tick,
                                double obs[FPV_ORACLE_YAW_OBS]);
uint32_t fpv_oracle_yaw_step_one(const fpv_oracle_cfg* cfg, double* y, double action, uint64_t drone_id,
                                 uint64_t agent_tick, double obs[FPV_ORACLE_YAW_OBS], double* reward);
void fpv_oracle_yaw_reset_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, double* obs,
                                const uint8_t* mask_or_null, uint64_t tick);
void fpv_oracle_yaw_step_batch(const fpv_oracle_cfg* cfg, int64_t n, double* state, const double* actions,
                               uint64_t agent_tick, double* obs, double* reward, uint8_t* terminated,
                               uint8_t* truncated, int64_t* led_count, int32_t* led_idx, uint32_t* led_info,
                               double* led_ret, double* led_obs);

int fpv_oracle_version(void);

#ifdef __cplusplus
}
#endif
#endif
