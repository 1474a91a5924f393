/*
 * rollout_host.c — the rollout loop of INTEGRATION.md driven from plain C through include/fpv_env.h only
 * (no Python, no torch): a randomly initialised MlpPolicy, n hover drones, T-step rollouts with VecNormalize
 * bookkeeping in the kernels, timeout bootstrap from the done ledger, GAE.  Prints env-steps/s and a few sanity
 * figures.  Build (examples/c_host/Makefile):  hipcc -x c rollout_host.c -I../../include -L<pkg> -lfpv_env
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fpv_env.h"

#define CK(call)                                                                         \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fpv_last_error()); exit(1); } \
    } while (0)
#define HK(call)                                                                      \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static void* dmalloc(size_t bytes) {
    void* p = NULL;
    HK(hipMalloc(&p, bytes));
    HK(hipMemset(p, 0, bytes));
    return p;
}

static float* upload_uniform(size_t n, float scale, unsigned* seed) {
    float* h = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        *seed = *seed * 1664525u + 1013904223u;
        h[i] = scale * ((float)(*seed >> 8) / 8388608.0f - 1.0f);
    }
    float* d = (float*)dmalloc(n * sizeof(float));
    HK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    free(h);
    return d;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    const int T = argc > 2 ? atoi(argv[2]) : 32, rollouts = argc > 3 ? atoi(argv[3]) : 16;
    /* 4th argument 2: policy forward and env step as two kernels per step; default 1: fpv_rollout_step, both in one kernel
     * (served when n is a multiple of 256) */
    const int one_kernel = (argc > 4 ? atoi(argv[4]) : 1) == 1 && n % 256 == 0;
    /* 5th argument: fpv_policy_set_gemm's mode — 1 (default): the 128-wide products as 6-term bf16 splits with fp32 accumulation, the
     * arithmetic the Python collector runs by default since round 5; 0: exact fp32-input MFMA */
    const int gemm = argc > 5 ? atoi(argv[5]) : 1;
    const float gamma = 0.99f, lam = 0.95f;
    if (fpv_version() != FPV_ABI_VERSION) {          /* struct layouts differ between versions: refuse another library build */
        fprintf(stderr, "libfpv_env.so is ABI version %d, this host was compiled against %d\n", fpv_version(), FPV_ABI_VERSION);
        return 3;
    }
    hipStream_t s;
    HK(hipSetDevice(0));
    HK(hipStreamCreate(&s));

    fpv_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.num_envs = n; cfg.device = 0; cfg.precision = FPV_F32; cfg.floor_check = 1; cfg.task = FPV_TASK_HOVER;
    cfg.noise_ratio = 0.02; cfg.seed = 1; cfg.ledger_capacity = 4 * n;
    fpv_env* env = NULL;
    CK(fpv_create(&cfg, &env));

    /* policy: 13 tensors in fpv_policy_load's order, small uniform weights (log_std = -1) */
    const size_t sizes[13] = {128 * 20, 128, 128 * 128, 128, 4 * 128, 4, 4, 128 * 20, 128, 128 * 128, 128, 128, 1};
    const float scales[13] = {0.2f, 0.f, 0.1f, 0.f, 0.01f, 0.f, 0.f, 0.2f, 0.f, 0.1f, 0.f, 0.1f, 0.f};
    const float* w[13];
    unsigned seed = 7u;
    for (int k = 0; k < 13; ++k) w[k] = upload_uniform(sizes[k], scales[k], &seed);
    {
        const float ls[4] = {-1.f, -1.f, -1.f, -1.f};
        HK(hipMemcpy((void*)w[6], ls, sizeof ls, hipMemcpyHostToDevice));
    }
    fpv_policy* pol = NULL;
    CK(fpv_policy_create(0, &pol));
    CK(fpv_policy_set_gemm(pol, gemm));
    CK(fpv_policy_load(pol, w, s));

    /* buffers */
    float *raw_obs = dmalloc(n * 20 * 4), *clipped = dmalloc(n * 16), *cur_obs = dmalloc(n * 20 * 4);
    float *buf_obs = dmalloc((size_t)T * n * 80), *buf_act = dmalloc((size_t)T * n * 16), *buf_val = dmalloc((size_t)T * n * 4);
    float *buf_logp = dmalloc((size_t)T * n * 4), *raw_rew = dmalloc((size_t)T * n * 4), *buf_rew = dmalloc((size_t)T * n * 4);
    float *adv = dmalloc((size_t)T * n * 4), *ret = dmalloc((size_t)T * n * 4), *last_val = dmalloc(n * 4), *ret_acc = dmalloc(n * 4);
    uint8_t *term = dmalloc(n), *trunc = dmalloc(n), *starts = dmalloc((size_t)(T + 1) * n);
    int64_t rows = 0;
    CK(fpv_partial_rows(env, &rows));
    if (one_kernel) rows = n / 256;                     /* fpv_rollout_step: one row of moment sums per 256-drone workgroup */
    double* part2[2] = {dmalloc((size_t)rows * 44 * 8), dmalloc((size_t)rows * 44 * 8)};   /* step t writes part2[t & 1] */
    double *partials = part2[0], *snap = dmalloc((size_t)T * 44 * 8), *adv_stats = dmalloc(3 * 8), *mean_std = dmalloc(2 * 8);
    double* stats[2] = {dmalloc(41 * 8), dmalloc(41 * 8)};
    double* rstats[2] = {dmalloc(3 * 8), dmalloc(3 * 8)};
    {   /* RunningMeanStd initial state: mean 0, var 1, count 1e-4 */
        double h[41] = {0};
        for (int c = 20; c < 40; ++c) h[c] = 1.0;
        h[40] = 1e-4;
        HK(hipMemcpy(stats[0], h, sizeof h, hipMemcpyHostToDevice));
        double r[3] = {0.0, 1.0, 1e-4};
        HK(hipMemcpy(rstats[0], r, sizeof r, hipMemcpyHostToDevice));
    }
    const int64_t cap = n;
    float *tobs = dmalloc(cap * 80), *tvals = dmalloc(cap * 4);
    int32_t *t_idx = dmalloc(cap * 4), *d_idx = dmalloc(cap * 4), *counters = dmalloc(16);
    double* sums = dmalloc(16);
    const uint64_t* tick_ptr = NULL;
    CK(fpv_tick_ptr(env, &tick_ptr));
    HK(hipMemset(starts + (size_t)T * n, 1, n));

    CK(fpv_reset(env, NULL, raw_obs, s));
    HK(hipStreamSynchronize(s));
    struct timespec t0, t1;
    for (int r = -2; r < rollouts; ++r) {           /* two warm-up rollouts */
        if (r == 0) { HK(hipStreamSynchronize(s)); clock_gettime(CLOCK_MONOTONIC, &t0); }
        CK(fpv_rollout_begin(env, starts, starts + (size_t)T * n, counters, sums, s));     /* ledger clear + episode-start carry + scan words, one kernel */
        for (int t = 0; t < T; ++t) {
            fpv_policy_io io;
            memset(&io, 0, sizeof io);
            io.n = n; io.obs = raw_obs; io.rms = stats[t ? (t - 1) & 1 : 0]; io.tick = tick_ptr;
            io.obs_n = buf_obs + (size_t)t * n * 20; io.action = buf_act + (size_t)t * n * 4; io.clipped = clipped;
            io.value = buf_val + (size_t)t * n; io.logp = buf_logp + (size_t)t * n;
            io.seed = 3; io.clip_obs = 10.f; io.eps_norm = 1e-8f;
            double* rows_in = one_kernel ? part2[(t - 1) & 1] : partials;      /* the table step t - 1 wrote */
            double* rows_out = one_kernel ? part2[t & 1] : partials;
            if (t > 0) {
                io.merge_partials = rows_in; io.merge_rows = rows; io.merge_count = (double)n;
                io.rms_out = stats[t & 1]; io.ret_rms = rstats[(t - 1) & 1]; io.ret_rms_out = rstats[t & 1];
                io.merge_snapshot = snap + (size_t)(t - 1) * 44;
            }
            fpv_step_extras ex;
            memset(&ex, 0, sizeof ex);
            ex.done = starts + (size_t)(t + 1) * n; ex.ret = ret_acc; ex.partials = rows_out; ex.gamma = gamma;
            if (one_kernel) {
                io.tick = NULL; io.clipped = NULL;                               /* the action stays on the chip */
                CK(fpv_rollout_step(env, pol, &io, raw_rew + (size_t)t * n, term, trunc, &ex, s));
            } else {
                CK(fpv_policy_forward(pol, &io, s));
                CK(fpv_step_fused(env, clipped, raw_obs, raw_rew + (size_t)t * n, term, trunc, &ex, s));
            }
        }
        CK(fpv_moments_merge_to(rows, 20, one_kernel ? part2[(T - 1) & 1] : partials, (double)n, stats[(T - 1) & 1],
                                rstats[(T - 1) & 1], stats[0], rstats[0], snap + (size_t)(T - 1) * 44, s));
        fpv_policy_io vio;
        memset(&vio, 0, sizeof vio);
        vio.n = n; vio.obs = raw_obs; vio.rms = stats[0]; vio.obs_n = cur_obs; vio.value = last_val;
        vio.deterministic = 1; vio.nets = 2; vio.clip_obs = 10.f; vio.eps_norm = 1e-8f;
        CK(fpv_policy_forward(pol, &vio, s));
        /* the ledger scan and the deferred reward scaling (fpv_reward_finalize) as one launch */
        CK(fpv_ledger_scan_finalize(env, snap, 44, T, 10.f, 1e-8f, tobs, t_idx, d_idx, cap, counters, sums, 1, NULL, raw_rew, snap, 44, 10.f,
                                    1e-8f, buf_rew, s));
        fpv_policy_io lio;
        memset(&lio, 0, sizeof lio);
        lio.n = cap; lio.obs = tobs; lio.value = tvals; lio.deterministic = 1; lio.nets = 2; lio.n_dev = counters;
        lio.clip_obs = 10.f; lio.eps_norm = 1e-8f;
        lio.scatter_t = t_idx; lio.scatter_d = d_idx; lio.scatter_out = buf_rew; lio.scatter_stride = n; lio.scatter_scale = gamma;
        CK(fpv_policy_forward(pol, &lio, s));        /* r[t][d] += gamma * V(terminal observation), in the launch that evaluates V */
        CK(fpv_gae(T, n, buf_rew, buf_val, starts, last_val, starts + (size_t)T * n, gamma, lam, adv, ret, adv_stats, NULL, s));
        CK(fpv_advantage_stats(adv_stats, 1, 3, mean_std, s));
    }
    HK(hipStreamSynchronize(s));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    int32_t hc[4];
    double hs[2], hm[2], hstats[41];
    HK(hipMemcpy(hc, counters, sizeof hc, hipMemcpyDeviceToHost));
    HK(hipMemcpy(hs, sums, sizeof hs, hipMemcpyDeviceToHost));
    HK(hipMemcpy(hm, mean_std, sizeof hm, hipMemcpyDeviceToHost));
    HK(hipMemcpy(hstats, stats[0], sizeof hstats, hipMemcpyDeviceToHost));
    const int ok = hc[3] == 0 && isfinite(hm[0]) && hm[1] > 0.0 && hstats[40] > (double)n * T * rollouts;
    printf("{\"drones\": %lld, \"T\": %d, \"rollouts\": %d, \"kernels_per_step\": %d, \"policy_gemm\": %d, \"env_steps_per_s\": %.4g, \"us_per_step\": %.2f, "
           "\"episodes_last_rollout\": %d, \"truncated_last_rollout\": %d, \"ledger_flags\": %d, \"ep_len_mean\": %.1f, "
           "\"adv_mean\": %.4g, \"adv_std\": %.4g, \"obs_count\": %.0f, \"ok\": %s}\n",
           (long long)n, T, rollouts, one_kernel ? 1 : 2, gemm, (double)n * T * rollouts / sec, sec * 1e6 / ((double)T * rollouts), hc[2], hc[0], hc[3],
           hc[1] ? hs[1] / hc[1] : 0.0, hm[0], hm[1], hstats[40], ok ? "true" : "false");
    CK(fpv_policy_destroy(pol));
    CK(fpv_destroy(env));
    return ok ? 0 : 2;
}
