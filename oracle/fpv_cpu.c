/*
 * fpv_cpu.c — the oracle behind the product's own C ABI.  TEST INFRASTRUCTURE ONLY (see fpv_oracle.h).
 *
 * SURVEY.md §8(b), last sentence of the C-ABI row: "a parallel fpv_cpu_* set with identical signatures over host
 * pointers (the oracle)".  Every function here has the signature of its namesake in include/fpv_env.h (fpv_X ->
 * fpv_cpu_X), takes HOST pointers where the product takes device pointers, ignores the stream argument and runs
 * oracle/fpv_oracle.c (double arithmetic whatever the handle's I/O precision; FPV_F32 handles convert at the boundary).
 * A parity test is then the same call sequence made twice — once against libfpv_env.so with device buffers, once
 * against this library with numpy buffers (tests/test_cpu_twin*.py).
 *
 * The done ledger keeps the product's partition (include/fpv_env.h, fpv_ledger): per 64-drone tile its own rows and
 * counter, rows appended in drone order, counters running past rows_per_tile.
 */
#include "../include/fpv_env.h"
#include "fpv_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TILE 64

struct fpv_env {
    fpv_config cfg;
    fpv_oracle_cfg ocfg;
    int64_t n, tiles, led_rows, led_cap;
    int state_words, obs_words, act_words;
    double* state;          /* [n][state_words], canonical order */
    uint64_t tick;
    uint32_t *led_count, *led_info, *led_tag;
    int32_t* led_idx;
    void *led_ret, *led_obs; /* float or double by precision */
    /* per-step scratch (double) */
    double *act, *obs, *rew;
    int64_t step_led_count;
    int32_t* s_idx;
    uint32_t* s_info;
    double *s_ret, *s_obs;
};

static _Thread_local char g_err[256] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char* fpv_cpu_last_error(void) { return g_err; }
int fpv_cpu_version(void) { return FPV_ABI_VERSION; }

int fpv_cpu_destroy(fpv_env* e) {
    if (!e) return FPV_OK;
    free(e->state); free(e->led_count); free(e->led_info); free(e->led_tag); free(e->led_idx); free(e->led_ret); free(e->led_obs);
    free(e->act); free(e->obs); free(e->rew); free(e->s_idx); free(e->s_info); free(e->s_ret); free(e->s_obs);
    free(e);
    return FPV_OK;
}

static int is_f64(const fpv_env* e) { return e->cfg.precision == FPV_F64; }
static void put(const fpv_env* e, void* dst, int64_t i, double v) {
    if (is_f64(e)) ((double*)dst)[i] = v;
    else ((float*)dst)[i] = (float)v;
}

int fpv_cpu_reset(fpv_env* e, const uint8_t* mask, void* obs, void* stream);

int fpv_cpu_create(const fpv_config* cfg, fpv_env** out) {
    if (!cfg || !out) return fail(FPV_ERR_ARG, "fpv_create: null argument");
    if (cfg->num_envs <= 0) return fail(FPV_ERR_ARG, "fpv_create: num_envs must be positive");
    if (cfg->precision != FPV_F32 && cfg->precision != FPV_F64) return fail(FPV_ERR_ARG, "fpv_create: unknown precision");
    if (cfg->task != FPV_TASK_HOVER && cfg->task != FPV_TASK_YAW) return fail(FPV_ERR_ARG, "fpv_create: unknown task");
    fpv_env* e = (fpv_env*)calloc(1, sizeof *e);
    if (!e) return fail(FPV_ERR_ALLOC, "fpv_create: out of memory");
    e->cfg = *cfg;
    e->n = cfg->num_envs;
    e->tiles = (e->n + TILE - 1) / TILE;
    const int yaw = cfg->task == FPV_TASK_YAW;
    e->state_words = yaw ? FPV_YAW_STATE_WORDS : FPV_STATE_WORDS;
    e->obs_words = yaw ? FPV_YAW_OBS_WORDS : FPV_OBS_WORDS;
    e->act_words = yaw ? FPV_YAW_ACT_WORDS : FPV_ACT_WORDS;
    e->ocfg.floor_check = cfg->floor_check;
    e->ocfg.domain_rand = cfg->domain_rand;
    e->ocfg.noise_ratio = cfg->noise_ratio;
    e->ocfg.seed = cfg->seed;
    e->ocfg.id_offset = cfg->id_offset;
    e->ocfg.detector = cfg->detector == FPV_DETECTOR_PIXEL ? 1 : 0;     /* (the oracle's switch: non-zero = pixel rules) */
    const int64_t want = cfg->ledger_capacity > 0 ? cfg->ledger_capacity : 2 * e->n;
    e->led_rows = (want + e->tiles - 1) / e->tiles;
    e->led_cap = e->led_rows * e->tiles;
    const size_t w = is_f64(e) ? 8 : 4;
    e->state = (double*)calloc((size_t)e->n * e->state_words, sizeof(double));
    e->led_count = (uint32_t*)calloc((size_t)e->tiles, sizeof(uint32_t));
    e->led_info = (uint32_t*)calloc((size_t)e->led_cap, sizeof(uint32_t));
    e->led_tag = (uint32_t*)calloc((size_t)e->led_cap, sizeof(uint32_t));
    e->led_idx = (int32_t*)calloc((size_t)e->led_cap, sizeof(int32_t));
    e->led_ret = calloc((size_t)e->led_cap, w);
    e->led_obs = calloc((size_t)e->led_cap * e->obs_words, w);
    e->act = (double*)calloc((size_t)e->n * e->act_words, sizeof(double));
    e->obs = (double*)calloc((size_t)e->n * e->obs_words, sizeof(double));
    e->rew = (double*)calloc((size_t)e->n, sizeof(double));
    e->s_idx = (int32_t*)calloc((size_t)e->n, sizeof(int32_t));
    e->s_info = (uint32_t*)calloc((size_t)e->n, sizeof(uint32_t));
    e->s_ret = (double*)calloc((size_t)e->n, sizeof(double));
    e->s_obs = (double*)calloc((size_t)e->n * e->obs_words, sizeof(double));
    if (!e->state || !e->led_count || !e->led_info || !e->led_tag || !e->led_idx || !e->led_ret || !e->led_obs || !e->act ||
        !e->obs || !e->rew || !e->s_idx || !e->s_info || !e->s_ret || !e->s_obs) {
        fpv_cpu_destroy(e);
        return fail(FPV_ERR_ALLOC, "fpv_create: out of memory");
    }
    *out = e;
    return fpv_cpu_reset(e, NULL, NULL, NULL);        /* a fresh handle holds reset drones, as the product's does */
}

int fpv_cpu_dims(fpv_env* e, int32_t* state_words, int32_t* obs_words, int32_t* act_words) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_dims: null handle");
    if (state_words) *state_words = e->state_words;
    if (obs_words) *obs_words = e->obs_words;
    if (act_words) *act_words = e->act_words;
    return FPV_OK;
}

/* hover.py:72-114 for the selected drones; prev_action survives (hover.py:101: the oracle's reset keeps it) */
int fpv_cpu_reset(fpv_env* e, const uint8_t* mask, void* obs, void* stream) {
    (void)stream;
    if (!e) return fail(FPV_ERR_ARG, "fpv_reset: null handle");
    if (e->cfg.task == FPV_TASK_YAW) fpv_oracle_yaw_reset_batch(&e->ocfg, e->n, e->state, e->obs, mask, e->tick);
    else fpv_oracle_reset_batch_d(e->n, e->state, e->obs, mask, e->ocfg.detector);
    if (obs)
        for (int64_t i = 0; i < e->n; ++i) {
            if (mask && !mask[i]) continue;
            for (int k = 0; k < e->obs_words; ++k) put(e, obs, i * e->obs_words + k, e->obs[i * e->obs_words + k]);
        }
    return FPV_OK;
}

static int step_common(fpv_env* e, void* obs, void* reward, uint8_t* term, uint8_t* trunc) {
    e->step_led_count = 0;
    if (e->cfg.task == FPV_TASK_YAW)
        fpv_oracle_yaw_step_batch(&e->ocfg, e->n, e->state, e->act, e->tick, e->obs, e->rew, term, trunc, &e->step_led_count,
                                  e->s_idx, e->s_info, e->s_ret, e->s_obs);
    else
        fpv_oracle_step_batch(&e->ocfg, e->n, e->state, e->act, e->tick, e->obs, e->rew, term, trunc, &e->step_led_count,
                              e->s_idx, e->s_info, e->s_ret, e->s_obs, 1);
    for (int64_t i = 0; i < e->n * e->obs_words; ++i) put(e, obs, i, e->obs[i]);
    for (int64_t i = 0; i < e->n; ++i) put(e, reward, i, e->rew[i]);
    /* this step's episode ends (ascending drone index) into the per-tile ledger */
    for (int64_t c = 0; c < e->step_led_count; ++c) {
        const int64_t tile = e->s_idx[c] / TILE;
        const uint32_t row = e->led_count[tile]++;
        if ((int64_t)row >= e->led_rows) continue;                       /* overflow rows are dropped, the count keeps counting */
        const int64_t slot = tile * e->led_rows + row;
        e->led_idx[slot] = e->s_idx[c];
        e->led_info[slot] = e->s_info[c];
        e->led_tag[slot] = (uint32_t)e->tick;
        put(e, e->led_ret, slot, e->s_ret[c]);
        for (int k = 0; k < e->obs_words; ++k) put(e, e->led_obs, slot * e->obs_words + k, e->s_obs[c * e->obs_words + k]);
    }
    e->tick += 1;
    return FPV_OK;
}

/* QuadXHoverEnv.step() (hover.py:334-358) for every drone, with VecEnv auto-reset */
int fpv_cpu_step(fpv_env* e, const float* actions, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, void* stream) {
    (void)stream;
    if (!e) return fail(FPV_ERR_ARG, "fpv_step: null handle");
    if (!actions || !obs || !reward || !terminated || !truncated) return fail(FPV_ERR_ARG, "fpv_step: null buffer");
    if (is_f64(e)) return fail(FPV_ERR_ARG, "fpv_step: handle precision does not match this entry point");
    for (int64_t i = 0; i < e->n * e->act_words; ++i) e->act[i] = (double)actions[i];
    return step_common(e, obs, reward, terminated, truncated);
}

int fpv_cpu_step_f64(fpv_env* e, const double* actions, double* obs, double* reward, uint8_t* terminated, uint8_t* truncated,
                     void* stream) {
    (void)stream;
    if (!e) return fail(FPV_ERR_ARG, "fpv_step: null handle");
    if (!actions || !obs || !reward || !terminated || !truncated) return fail(FPV_ERR_ARG, "fpv_step: null buffer");
    if (!is_f64(e)) return fail(FPV_ERR_ARG, "fpv_step: handle precision does not match this entry point");
    memcpy(e->act, actions, (size_t)e->n * e->act_words * sizeof(double));
    return step_common(e, obs, reward, terminated, truncated);
}

/* canonical state blobs [num_envs][state words] in the handle's precision (docs/SPEC.md §2) */
int fpv_cpu_get_state(fpv_env* e, void* state_out, void* stream) {
    (void)stream;
    if (!e || !state_out) return fail(FPV_ERR_ARG, "fpv_get_state: null argument");
    for (int64_t i = 0; i < e->n * e->state_words; ++i) put(e, state_out, i, e->state[i]);
    return FPV_OK;
}
int fpv_cpu_set_state(fpv_env* e, const void* state_in, void* stream) {
    (void)stream;
    if (!e || !state_in) return fail(FPV_ERR_ARG, "fpv_set_state: null argument");
    for (int64_t i = 0; i < e->n * e->state_words; ++i)
        e->state[i] = is_f64(e) ? ((const double*)state_in)[i] : (double)((const float*)state_in)[i];
    return FPV_OK;
}

int fpv_cpu_set_tick(fpv_env* e, uint64_t tick, void* stream) {
    (void)stream;
    if (!e) return fail(FPV_ERR_ARG, "fpv_set_tick: null handle");
    e->tick = tick;
    return FPV_OK;
}
int fpv_cpu_get_tick(fpv_env* e, uint64_t* tick_host, void* stream) {
    (void)stream;
    if (!e || !tick_host) return fail(FPV_ERR_ARG, "fpv_get_tick: null argument");
    *tick_host = e->tick;
    return FPV_OK;
}

int fpv_cpu_ledger_view(fpv_env* e, fpv_ledger* out) {
    if (!e || !out) return fail(FPV_ERR_ARG, "fpv_ledger_view: null argument");
    out->capacity = e->led_cap;
    out->rows_per_tile = e->led_rows;
    out->count = e->led_count; out->idx = e->led_idx; out->info = e->led_info; out->tag = e->led_tag;
    out->ep_return = e->led_ret; out->obs = e->led_obs;
    return FPV_OK;
}
int fpv_cpu_ledger_clear(fpv_env* e, void* stream) {
    (void)stream;
    if (!e) return fail(FPV_ERR_ARG, "fpv_ledger_clear: null handle");
    memset(e->led_count, 0, (size_t)e->tiles * sizeof(uint32_t));
    return FPV_OK;
}

/* The done ledger as one dense block (include/fpv_env.h, fpv_ledger_pack): header {appended, packed, W, held, 0, 0, 0, 0}, then
 * {idx, info, tag, ep_return, obs[O]} per row in (tile, row) order; with obs_rms the terminal observation normalised in float as
 * fpv_normalize does: clip((x - (float)mean) * (float)(1 / sqrt(var + eps)), +-clip). */
int fpv_cpu_ledger_pack(fpv_env* e, const double* obs_rms, float clip_obs, float eps_norm, uint32_t* out, int64_t max_rows, void* stream) {
    (void)stream;
    if (!e || !out) return fail(FPV_ERR_ARG, "fpv_ledger_pack: null argument");
    if (e->cfg.precision != FPV_F32) return fail(FPV_ERR_UNSUPPORTED, "fpv_ledger_pack: float handles only");
    if (max_rows < 0) return fail(FPV_ERR_ARG, "fpv_ledger_pack: max_rows >= 0, out 4-byte aligned");
    const int O = e->obs_words, W = 4 + O;
    uint32_t appended = 0, held = 0;
    int64_t k = 0;
    for (int64_t t = 0; t < e->tiles; ++t) {
        const uint32_t c = e->led_count[t] < (uint32_t)e->led_rows ? e->led_count[t] : (uint32_t)e->led_rows;
        appended += e->led_count[t];
        held += c;
        for (uint32_t r = 0; r < c; ++r, ++k) {
            if (k >= max_rows) continue;
            const int64_t slot = t * e->led_rows + r;
            uint32_t* row = out + 8 + k * W;
            row[0] = (uint32_t)e->led_idx[slot];
            row[1] = e->led_info[slot];
            row[2] = e->led_tag[slot];
            memcpy(row + 3, (const float*)e->led_ret + slot, 4);
            for (int j = 0; j < O; ++j) {
                float x = ((const float*)e->led_obs)[slot * O + j];
                if (obs_rms) {
                    const float mu = (float)obs_rms[j], inv = (float)(1.0 / sqrt(obs_rms[O + j] + (double)eps_norm));
                    x = fminf(fmaxf((x - mu) * inv, -clip_obs), clip_obs);
                }
                memcpy(row + 4 + j, &x, 4);
            }
        }
    }
    out[0] = appended;
    out[1] = (uint32_t)(held < max_rows ? held : max_rows);
    out[2] = (uint32_t)W;
    out[3] = held;
    out[4] = out[5] = out[6] = out[7] = 0u;
    return FPV_OK;
}

/* SB3 RolloutBuffer.compute_returns_and_advantage; stats = {count, sum A, sum A^2}, mean_std = {mean, population std} */
int fpv_cpu_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
                const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* advantages, float* returns,
                double* stats, double* mean_std, void* stream) {
    (void)stream;
    if (T <= 0 || n <= 0) return fail(FPV_ERR_ARG, "fpv_gae: T and n must be positive");
    if (mean_std && !stats) return fail(FPV_ERR_ARG, "fpv_gae: mean_std needs stats");
    if (!rewards || !values || !episode_starts || !last_values || !last_dones || !advantages || !returns)
        return fail(FPV_ERR_ARG, "fpv_gae: null buffer");
    const size_t tn = (size_t)T * (size_t)n;
    double* buf = (double*)malloc((4 * tn + (size_t)n) * sizeof(double));
    if (!buf) return fail(FPV_ERR_ALLOC, "fpv_gae: out of memory");
    double *r = buf, *v = buf + tn, *a = buf + 2 * tn, *ret = buf + 3 * tn, *lv = buf + 4 * tn;
    for (size_t i = 0; i < tn; ++i) { r[i] = rewards[i]; v[i] = values[i]; }
    for (int64_t i = 0; i < n; ++i) lv[i] = last_values[i];
    fpv_oracle_gae(T, n, r, v, episode_starts, lv, last_dones, (double)gamma, (double)lam, a, ret);
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = 0; i < tn; ++i) {
        advantages[i] = (float)a[i];
        returns[i] = (float)ret[i];
        s1 += (double)advantages[i];
        s2 += (double)advantages[i] * (double)advantages[i];
    }
    if (stats) { stats[0] = (double)tn; stats[1] = s1; stats[2] = s2; }
    if (mean_std) {
        const double m = s1 / (double)tn;
        double var = s2 / (double)tn - m * m;
        if (var < 0.0) var = 0.0;
        mean_std[0] = m;
        mean_std[1] = sqrt(var);
    }
    free(buf);
    return FPV_OK;
}
