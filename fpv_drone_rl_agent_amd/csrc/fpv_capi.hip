// fpv_capi.hip — host side of the C ABI declared in include/fpv_env.h.
// Owns device allocations for the persistent drone state, the reset template, the done
// ledger and the tick; launches the kernels of fpv_step.hip / fpv_rollout.hip on the
// caller's stream.  No allocation and no host synchronisation inside fpv_step.
#include "../../include/fpv_env.h"
#include "../../include/fpv_env_debug.h"
#ifdef FPV_EXPERIMENTS
#include "../../include/fpv_env_experiments.h"
#endif
#include "fpv_dynamics.h"
#include "fpv_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

using namespace fpv;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define FPV_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(FPV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

struct fpv_env {
    fpv_config cfg;
    int64_t tiles;
    int granules, state_words, obs_words, act_words;  // by task
    size_t word;         // sizeof(real)
    void* state;         // [tiles][9][64] granules
    void* tmpl;          // 56 reals
    uint64_t* tick;      // device [tiles]
    uint32_t* merge_ticket;   // 16 device words, zero between launches: [0] the in-launch statistics merge, [4] fpv_ledger_scan's last-workgroup ticket
    uint32_t* rollout_sync;   // barrier words of fpv_rollout_steps (zeroed on the stream before every launch)
    uint64_t* stamps;         // fpv_debug_rollout_stamps: diagnostic buffer of this handle's rollout kernels, or null
    uint64_t timeout_ticks;   // fpv_debug_rollout_timeout: 0 = the default bound of fpv_rollout_steps' grid-wide waits
    uint32_t* led_count;
    int32_t* led_idx;
    uint32_t* led_info;
    uint32_t* led_tag;
    void* led_ret;
    void* led_obs;
    int64_t led_cap;     // tiles * led_rows
    int64_t led_rows;    // rows per tile
    bool led_owned;
};

struct fpv_policy {
    int device;
    int cus;
    int gemm;               // 0: exact fp32 MFMA, 1: 6-term bf16 split MFMA, 2: exact fp32 MFMA on 16x16x4 tiles, 16-wave workgroups (fpv_policy_w16.hip)
    int in_dim, act_dim;    // 20 / 4 (hover) unless fpv_policy_set_dims says otherwise
    float* image;
    float* image16;         // fpv_policy_w16.hip
    unsigned char* split_image;
};

// per-device reduction scratch for the handle-less rollout primitives
static std::mutex g_scratch_mu;
static double* g_scratch[64] = {};
static int scratch_for_current_device(double** out) {
    int dev = 0;
    FPV_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FPV_ERR_ARG, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch[dev]) {
        FPV_HIP(hipMalloc(&g_scratch[dev], sizeof(double) * kScratchDoubles));
        FPV_HIP(hipMemset(g_scratch[dev], 0, sizeof(double) * kScratchDoubles));   // holds the tickets of in-launch reductions
    }
    *out = g_scratch[dev];
    return FPV_OK;
}

template <typename T>
static int step_impl(fpv_env* e, const T* actions, T* obs, T* reward, uint8_t* term, uint8_t* trunc, void* stream,
                     const fpv_step_extras* ex = nullptr) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_step: null handle");
    if (!actions || !obs || !reward || !term || !trunc) return fail(FPV_ERR_ARG, "fpv_step: null buffer");
    if ((e->cfg.precision == FPV_F64) != (sizeof(T) == 8))
        return fail(FPV_ERR_ARG, "fpv_step: handle precision does not match this entry point");
    if (((uintptr_t)obs | (e->act_words == 4 ? (uintptr_t)actions : 0)) & (sizeof(T) * 4 - 1))
        return fail(FPV_ERR_ARG, "fpv_step: actions/obs must be %zu-byte aligned", sizeof(T) * 4);

    StepArgs<T> a;
    a.state = (Vec4<T>*)e->state;
    a.actions = actions; a.obs = obs; a.reward = reward; a.terminated = term; a.truncated = trunc;
    a.tick = e->tick;
    a.tmpl = (const T*)e->tmpl;
    a.led_count = e->led_count; a.led_idx = e->led_idx; a.led_info = e->led_info; a.led_tag = e->led_tag;
    a.led_ret = (T*)e->led_ret; a.led_obs = (T*)e->led_obs; a.led_rows = e->led_rows;
    a.n = e->cfg.num_envs;
    a.seed = e->cfg.seed;
    a.id_offset = e->cfg.id_offset;
    a.sigma = (T)e->cfg.noise_ratio;
    a.floor_check = e->cfg.floor_check;
    a.domain_rand = e->cfg.domain_rand;
    a.detector = e->cfg.detector == FPV_DETECTOR_PIXEL ? 1 : 0;   // kernel-side switch: non-zero = pixel rules
    a.done = ex ? ex->done : nullptr;
    a.ret = ex ? ex->ret : nullptr;
    a.partials = ex ? ex->partials : nullptr;
    a.gamma = ex ? ex->gamma : 0.0f;
    const bool want_merge = ex && (ex->obs_rms || ex->ret_rms);
    if (want_merge && !ex->partials) return fail(FPV_ERR_ARG, "fpv_step_fused: obs_rms / ret_rms need partials");
    a.mg_ticket = want_merge ? e->merge_ticket : nullptr;
    a.mg_obs_rms = want_merge ? ex->obs_rms : nullptr;
    a.mg_ret_rms = want_merge ? ex->ret_rms : nullptr;
    a.mg_snapshot = want_merge ? ex->snapshot : nullptr;
    a.mg_count = (double)e->cfg.num_envs;
    bool merged = false;
    if (e->cfg.task == FPV_TASK_YAW) FPV_HIP(launch_yaw_step<T>(a, e->cfg.noise_ratio != 0.0, (hipStream_t)stream, &merged));
    else FPV_HIP(launch_step<T>(a, e->cfg.noise_ratio != 0.0, (hipStream_t)stream, &merged));
    if (want_merge && !merged) {
        // launch shapes without the in-kernel reducer (one tile per workgroup, the wave-pair kernel): the merge kernel
        int64_t rows = e->cfg.task == FPV_TASK_YAW ? yaw_partial_rows(e->cfg.num_envs)
                                                   : step_partial_rows(e->cfg.num_envs, e->cfg.noise_ratio != 0.0, e->cfg.detector == FPV_DETECTOR_PIXEL);
        double* scratch = nullptr;
        if (int rc = scratch_for_current_device(&scratch)) return rc;
        FPV_HIP(launch_moments_merge_tiles(rows, ex->partials, (double)e->cfg.num_envs, nullptr, nullptr, ex->obs_rms, ex->ret_rms,
                                           ex->snapshot, scratch, e->obs_words, (hipStream_t)stream));
    }
    return FPV_OK;
}

__global__ void fpv_noop_kernel() {}

// the library is compiled with -fvisibility=hidden: only the C ABI of include/*.h is exported
#pragma GCC visibility push(default)
extern "C" {

int fpv_set_pair_threshold(int64_t max_drones) {
    fpv::g_pair_max_tiles = max_drones <= 0 ? 0 : (max_drones + kTile - 1) / kTile;
    return FPV_OK;
}

int fpv_set_quad_range(int64_t min_drones, int64_t max_drones) {
    fpv::g_quad_min_tiles = min_drones <= 0 ? 0 : (min_drones + kTile - 1) / kTile;
    fpv::g_quad_max_tiles = max_drones <= 0 ? 0 : (max_drones + kTile - 1) / kTile;
    return FPV_OK;
}

int fpv_debug_rollout_stamps(fpv_env* e, uint64_t* device_buffer) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_debug_rollout_stamps: null handle");
    e->stamps = device_buffer;
    return FPV_OK;
}

int fpv_debug_rollout_timeout(fpv_env* e, uint64_t ticks) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_debug_rollout_timeout: null handle");
    e->timeout_ticks = ticks;
    return FPV_OK;
}

__global__ void fpv_atomic_inc_kernel(uint32_t* w, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&w[i], 1u);
}

int fpv_debug_memset(void* dst, int32_t value, int64_t bytes, void* stream) {
    if (!dst || bytes <= 0) return fail(FPV_ERR_ARG, "fpv_debug_memset: null destination or no bytes");
    FPV_HIP(hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_debug_atomic_inc(uint32_t* words, int64_t n, void* stream) {
    if (!words || n <= 0) return fail(FPV_ERR_ARG, "fpv_debug_atomic_inc: null words or n <= 0");
    hipLaunchKernelGGL(fpv_atomic_inc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, words, n);
    FPV_HIP(hipGetLastError());
    return FPV_OK;
}

int fpv_noop(void* stream) {
    hipLaunchKernelGGL(fpv_noop_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    FPV_HIP(hipGetLastError());
    return FPV_OK;
}

const char* fpv_last_error(void) { return g_err; }
int fpv_version(void) { return FPV_ABI_VERSION; }

int fpv_create(const fpv_config* cfg, fpv_env** out) {
    if (!cfg || !out) return fail(FPV_ERR_ARG, "fpv_create: null argument");
    if (cfg->num_envs <= 0) return fail(FPV_ERR_ARG, "fpv_create: num_envs must be positive");
    if (cfg->num_envs > (int64_t)1 << 31) return fail(FPV_ERR_ARG, "fpv_create: num_envs too large");
    if (cfg->precision != FPV_F32 && cfg->precision != FPV_F64)
        return fail(FPV_ERR_ARG, "fpv_create: unknown precision %d", cfg->precision);
    if (cfg->noise_ratio < 0.0) return fail(FPV_ERR_ARG, "fpv_create: negative noise_ratio");
    if (cfg->task != FPV_TASK_HOVER && cfg->task != FPV_TASK_YAW) return fail(FPV_ERR_ARG, "fpv_create: unknown task %d", cfg->task);
    if (cfg->detector != FPV_DETECTOR_ANALYTIC && cfg->detector != FPV_DETECTOR_PIXEL)
        return fail(FPV_ERR_ARG, "fpv_create: unknown detector %d", cfg->detector);
    int ndev = 0;
    FPV_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(FPV_ERR_ARG, "fpv_create: device %d not present (%d visible)", cfg->device, ndev);
    FPV_HIP(hipSetDevice(cfg->device));
    fpv_env* e = new (std::nothrow) fpv_env();
    if (!e) return fail(FPV_ERR_ALLOC, "fpv_create: host allocation failed");
    e->cfg = *cfg;
    e->tiles = (cfg->num_envs + kTile - 1) / kTile;
    e->word = cfg->precision == FPV_F64 ? sizeof(double) : sizeof(float);
    const bool yaw = cfg->task == FPV_TASK_YAW;
    e->granules = yaw ? kYawGranules : kGranules;
    e->state_words = yaw ? kYawStateWords : kStateWords;
    e->obs_words = yaw ? kYawObsWords : kObsWords;
    e->act_words = yaw ? 1 : 4;
    {
        const int64_t want = cfg->ledger_capacity > 0 ? cfg->ledger_capacity : 2 * cfg->num_envs;
        e->led_rows = (want + e->tiles - 1) / e->tiles;
        e->led_cap = e->led_rows * e->tiles;
    }
    e->led_owned = true;
    hipError_t err = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(p, bytes);
    };
    alloc(&e->state, (size_t)e->tiles * e->granules * kTile * 4 * e->word);
    alloc(&e->tmpl, (size_t)(kStateWords + kObsWords) * e->word);
    alloc((void**)&e->tick, (size_t)e->tiles * sizeof(uint64_t));
    alloc((void**)&e->merge_ticket, 64);
    alloc((void**)&e->rollout_sync, sizeof(uint32_t) * (size_t)(rollout_sync_words() > 0 ? rollout_sync_words() : 16));
    alloc((void**)&e->led_count, (size_t)e->tiles * sizeof(uint32_t));
    alloc((void**)&e->led_idx, (size_t)e->led_cap * sizeof(int32_t));
    alloc((void**)&e->led_info, (size_t)e->led_cap * sizeof(uint32_t));
    alloc((void**)&e->led_tag, (size_t)e->led_cap * sizeof(uint32_t));
    alloc(&e->led_ret, (size_t)e->led_cap * e->word);
    alloc(&e->led_obs, (size_t)e->led_cap * kObsWords * e->word);
    if (err == hipSuccess) err = hipMemset(e->tick, 0, (size_t)e->tiles * sizeof(uint64_t));
    if (err == hipSuccess) err = hipMemset(e->merge_ticket, 0, 64);
    if (err == hipSuccess) err = hipMemset(e->led_count, 0, (size_t)e->tiles * sizeof(uint32_t));
    if (err == hipSuccess) {
        if (yaw)
            err = cfg->precision == FPV_F64
                      ? launch_yaw_reset<double>(e->state, e->tick, nullptr, nullptr, cfg->num_envs, cfg->seed, cfg->id_offset,
                                                 cfg->domain_rand, 1, nullptr)
                      : launch_yaw_reset<float>(e->state, e->tick, nullptr, nullptr, cfg->num_envs, cfg->seed, cfg->id_offset,
                                                cfg->domain_rand, 1, nullptr);
        else
            err = cfg->precision == FPV_F64 ? launch_init<double>(e->state, e->tmpl, cfg->num_envs, cfg->detector == FPV_DETECTOR_PIXEL, nullptr)
                                            : launch_init<float>(e->state, e->tmpl, cfg->num_envs, cfg->detector == FPV_DETECTOR_PIXEL, nullptr);
    }
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        fpv_destroy(e);
        return fail(FPV_ERR_HIP, "fpv_create: %s", hipGetErrorString(err));
    }
    *out = e;
    return FPV_OK;
}

int fpv_destroy(fpv_env* e) {
    if (!e) return FPV_OK;
    void* own[] = {e->state, e->tmpl, e->tick, e->merge_ticket, e->rollout_sync};
    for (void* p : own)
        if (p) (void)hipFree(p);
    if (e->led_owned) {
        void* led[] = {e->led_count, e->led_idx, e->led_info, e->led_tag, e->led_ret, e->led_obs};
        for (void* p : led)
            if (p) (void)hipFree(p);
    }
    delete e;
    return FPV_OK;
}

int fpv_reset(fpv_env* e, const uint8_t* mask, void* obs, void* stream) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_reset: null handle");
    hipStream_t s = (hipStream_t)stream;
    if (e->cfg.task == FPV_TASK_YAW) {
        FPV_HIP(e->cfg.precision == FPV_F64
                    ? launch_yaw_reset<double>(e->state, e->tick, mask, obs, e->cfg.num_envs, e->cfg.seed, e->cfg.id_offset,
                                               e->cfg.domain_rand, 0, s)
                    : launch_yaw_reset<float>(e->state, e->tick, mask, obs, e->cfg.num_envs, e->cfg.seed, e->cfg.id_offset,
                                              e->cfg.domain_rand, 0, s));
        return FPV_OK;
    }
    FPV_HIP(e->cfg.precision == FPV_F64 ? launch_reset<double>(e->state, e->tmpl, mask, obs, e->cfg.num_envs, s)
                                        : launch_reset<float>(e->state, e->tmpl, mask, obs, e->cfg.num_envs, s));
    return FPV_OK;
}

int fpv_step(fpv_env* e, const float* actions, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
             void* stream) {
    return step_impl<float>(e, actions, obs, reward, terminated, truncated, stream);
}

int fpv_step_fused(fpv_env* e, const float* actions, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
                   const fpv_step_extras* extras, void* stream) {
    if (e && e->cfg.precision != FPV_F32) return fail(FPV_ERR_UNSUPPORTED, "fpv_step_fused: fp32 handles only");
    return step_impl<float>(e, actions, obs, reward, terminated, truncated, stream, extras);
}

int fpv_moments_merge(int64_t tiles, const double* partials, double batch_count, double* obs_rms, double* ret_rms,
                      double* snapshot, void* stream) {
    if (tiles <= 0 || !partials || batch_count <= 0.0) return fail(FPV_ERR_ARG, "fpv_moments_merge: bad arguments");
    double* scratch = nullptr;
    if (int rc = scratch_for_current_device(&scratch)) return rc;
    FPV_HIP(launch_moments_merge_tiles(tiles, partials, batch_count, nullptr, nullptr, obs_rms, ret_rms, snapshot, scratch,
                                       20, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_moments_merge_to(int64_t tiles, int32_t obs_words, const double* partials, double batch_count,
                         const double* obs_rms_in, const double* ret_rms_in, double* obs_rms_out, double* ret_rms_out,
                         double* snapshot, void* stream) {
    if (tiles <= 0 || !partials || batch_count <= 0.0) return fail(FPV_ERR_ARG, "fpv_moments_merge_to: bad arguments");
    if (obs_words < 1 || obs_words > 20) return fail(FPV_ERR_ARG, "fpv_moments_merge_to: obs_words must be 1..20");
    if ((obs_rms_out && !obs_rms_in) || (ret_rms_out && !ret_rms_in))
        return fail(FPV_ERR_ARG, "fpv_moments_merge_to: an output statistic needs its input");
    double* scratch = nullptr;
    if (int rc = scratch_for_current_device(&scratch)) return rc;
    FPV_HIP(launch_moments_merge_tiles(tiles, partials, batch_count, obs_rms_in, ret_rms_in, obs_rms_out, ret_rms_out,
                                       snapshot, scratch, obs_words, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_reward_finalize(int64_t T, int64_t n, const float* raw, const double* snapshots, float clip, float eps,
                        float* out, void* stream) {
    if (T <= 0 || n <= 0 || !raw || !snapshots || !out) return fail(FPV_ERR_ARG, "fpv_reward_finalize: bad arguments");
    FPV_HIP(launch_reward_finalize(T, n, raw, snapshots, clip, eps, out, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_partial_rows(fpv_env* e, int64_t* rows) {
    if (!e || !rows) return fail(FPV_ERR_ARG, "fpv_partial_rows: null argument");
    if (e->cfg.precision != 0) return fail(FPV_ERR_UNSUPPORTED, "fpv_partial_rows: fpv_step_fused serves fp32 handles");
    *rows = e->cfg.task == FPV_TASK_YAW ? yaw_partial_rows(e->cfg.num_envs)
                                        : step_partial_rows(e->cfg.num_envs, e->cfg.noise_ratio != 0.0, e->cfg.detector == FPV_DETECTOR_PIXEL);
    return FPV_OK;
}

int fpv_dims(fpv_env* e, int32_t* state_words, int32_t* obs_words, int32_t* act_words) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_dims: null handle");
    if (state_words) *state_words = e->state_words;
    if (obs_words) *obs_words = e->obs_words;
    if (act_words) *act_words = e->act_words;
    return FPV_OK;
}

int fpv_tick_ptr(fpv_env* e, const uint64_t** out) {
    if (!e || !out) return fail(FPV_ERR_ARG, "fpv_tick_ptr: null argument");
    *out = e->tick;
    return FPV_OK;
}

int fpv_step_f64(fpv_env* e, const double* actions, double* obs, double* reward, uint8_t* terminated,
                 uint8_t* truncated, void* stream) {
    return step_impl<double>(e, actions, obs, reward, terminated, truncated, stream);
}

int fpv_get_state(fpv_env* e, void* state_out, void* stream) {
    if (!e || !state_out) return fail(FPV_ERR_ARG, "fpv_get_state: null argument");
    if (e->cfg.task == FPV_TASK_YAW) {
        FPV_HIP(e->cfg.precision == FPV_F64 ? launch_yaw_convert<double>(e->state, state_out, e->cfg.num_envs, true, (hipStream_t)stream)
                                            : launch_yaw_convert<float>(e->state, state_out, e->cfg.num_envs, true, (hipStream_t)stream));
        return FPV_OK;
    }
    FPV_HIP(e->cfg.precision == FPV_F64
                ? launch_convert<double>(e->state, state_out, e->cfg.num_envs, true, (hipStream_t)stream)
                : launch_convert<float>(e->state, state_out, e->cfg.num_envs, true, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_set_state(fpv_env* e, const void* state_in, void* stream) {
    if (!e || !state_in) return fail(FPV_ERR_ARG, "fpv_set_state: null argument");
    if (e->cfg.task == FPV_TASK_YAW) {
        FPV_HIP(e->cfg.precision == FPV_F64
                    ? launch_yaw_convert<double>(e->state, (void*)state_in, e->cfg.num_envs, false, (hipStream_t)stream)
                    : launch_yaw_convert<float>(e->state, (void*)state_in, e->cfg.num_envs, false, (hipStream_t)stream));
        return FPV_OK;
    }
    FPV_HIP(e->cfg.precision == FPV_F64
                ? launch_convert<double>(e->state, (void*)state_in, e->cfg.num_envs, false, (hipStream_t)stream)
                : launch_convert<float>(e->state, (void*)state_in, e->cfg.num_envs, false, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_set_tick(fpv_env* e, uint64_t tick, void* stream) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_set_tick: null handle");
    FPV_HIP(launch_tick_fill(e->tick, e->tiles, tick, (hipStream_t)stream));
    FPV_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FPV_OK;
}

int fpv_get_tick(fpv_env* e, uint64_t* tick_host, void* stream) {
    if (!e || !tick_host) return fail(FPV_ERR_ARG, "fpv_get_tick: null argument");
    FPV_HIP(hipMemcpyAsync(tick_host, e->tick, sizeof *tick_host, hipMemcpyDeviceToHost, (hipStream_t)stream));
    FPV_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FPV_OK;
}

int fpv_ledger_view(fpv_env* e, fpv_ledger* out) {
    if (!e || !out) return fail(FPV_ERR_ARG, "fpv_ledger_view: null argument");
    out->capacity = e->led_cap;
    out->rows_per_tile = e->led_rows;
    out->count = e->led_count;
    out->idx = e->led_idx;
    out->info = e->led_info;
    out->tag = e->led_tag;
    out->ep_return = e->led_ret;
    out->obs = e->led_obs;
    return FPV_OK;
}

int fpv_ledger_bind(fpv_env* e, const fpv_ledger* b) {
    if (!e || !b) return fail(FPV_ERR_ARG, "fpv_ledger_bind: null argument");
    if (b->capacity <= 0 || !b->count || !b->idx || !b->info || !b->tag || !b->ep_return || !b->obs)
        return fail(FPV_ERR_ARG, "fpv_ledger_bind: every buffer and a positive capacity are required");
    if (b->rows_per_tile <= 0 || b->capacity != b->rows_per_tile * e->tiles)
        return fail(FPV_ERR_ARG, "fpv_ledger_bind: capacity must be rows_per_tile * %lld tiles", (long long)e->tiles);
    if (e->led_owned) {
        FPV_HIP(hipDeviceSynchronize());
        void* led[] = {e->led_count, e->led_idx, e->led_info, e->led_tag, e->led_ret, e->led_obs};
        for (void* p : led)
            if (p) (void)hipFree(p);
        e->led_owned = false;
    }
    e->led_cap = b->capacity;
    e->led_rows = b->rows_per_tile;
    e->led_count = b->count; e->led_idx = b->idx; e->led_info = b->info; e->led_tag = b->tag;
    e->led_ret = b->ep_return; e->led_obs = b->obs;
    return FPV_OK;
}

int fpv_ledger_clear(fpv_env* e, void* stream) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_ledger_clear: null handle");
    FPV_HIP(launch_zero_words(e->led_count, e->tiles, (hipStream_t)stream));   // a kernel: safe inside a captured graph
    return FPV_OK;
}

int fpv_ledger_pack(fpv_env* e, const double* obs_rms, float clip_obs, float eps_norm, uint32_t* out, int64_t max_rows, void* stream) {
    if (!e || !out) return fail(FPV_ERR_ARG, "fpv_ledger_pack: null argument");
    if (e->cfg.precision != FPV_F32) return fail(FPV_ERR_UNSUPPORTED, "fpv_ledger_pack: float handles only");
    if (max_rows < 0 || ((uintptr_t)out & 3)) return fail(FPV_ERR_ARG, "fpv_ledger_pack: max_rows >= 0, out 4-byte aligned");
    if (e->obs_words > 32) return fail(FPV_ERR_UNSUPPORTED, "fpv_ledger_pack: at most 32 observation words");
    FPV_HIP(launch_ledger_pack(e->led_count, e->led_idx, e->led_info, e->led_tag, static_cast<const float*>(e->led_ret),
                               static_cast<const float*>(e->led_obs), e->tiles, (int)e->led_rows, e->obs_words, obs_rms, clip_obs,
                               eps_norm, out, max_rows, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_rollout_begin(fpv_env* e, uint8_t* starts_dst, const uint8_t* starts_src, int32_t* scan_counters, double* scan_sums,
                      void* stream) {
    if (!e) return fail(FPV_ERR_ARG, "fpv_rollout_begin: null handle");
    if ((starts_dst == nullptr) != (starts_src == nullptr)) return fail(FPV_ERR_ARG, "fpv_rollout_begin: give both rows or neither");
    if ((scan_counters == nullptr) != (scan_sums == nullptr))
        return fail(FPV_ERR_ARG, "fpv_rollout_begin: give both scan_counters and scan_sums or neither");
    if (((uintptr_t)starts_dst | (uintptr_t)starts_src) & 15) return fail(FPV_ERR_ARG, "fpv_rollout_begin: rows must be 16-byte aligned");
    FPV_HIP(launch_rollout_begin(e->led_count, e->tiles, starts_dst, starts_src, e->cfg.num_envs, scan_counters, scan_sums,
                                 (hipStream_t)stream));
    return FPV_OK;
}

int fpv_advantage_stats(const double* gathered, int32_t world, int32_t stride, double* mean_std, void* stream) {
    if (!gathered || !mean_std || world <= 0 || stride < 3) return fail(FPV_ERR_ARG, "fpv_advantage_stats: bad arguments");
    FPV_HIP(launch_advantage_stats(gathered, world, stride, mean_std, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_rank_merge(const double* gathered, int32_t world, int32_t stride, int32_t obs_words, const double* base_obs_rms,
                   const double* base_ret_rms, double* obs_rms, double* ret_rms, void* stream) {
    if (!gathered || world <= 0 || obs_words <= 0 || obs_words > 32) return fail(FPV_ERR_ARG, "fpv_rank_merge: bad arguments");
    if (stride < 3 + 2 * obs_words + 1 + 3) return fail(FPV_ERR_ARG, "fpv_rank_merge: rows hold {adv[3], obs_rms[2 O + 1], ret_rms[3]}");
    if ((base_obs_rms == nullptr) != (obs_rms == nullptr) || (base_ret_rms == nullptr) != (ret_rms == nullptr))
        return fail(FPV_ERR_ARG, "fpv_rank_merge: give a statistics buffer together with its base, or neither");
    FPV_HIP(launch_rank_merge(gathered, world, stride, obs_words, base_obs_rms, base_ret_rms, obs_rms, ret_rms, (hipStream_t)stream));
    return FPV_OK;
}

static int ledger_scan_impl(const char* who, fpv_env* e, const double* snapshots, int64_t snapshot_stride, int32_t T, float clip_obs,
                            float eps_norm, float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                            int32_t counters_are_zero, void* status_out, const float* raw, const double* rew_snapshots,
                            int64_t rew_stride, float clip_reward, float eps_reward, float* rewards, void* stream) {
    if (!e || !tobs || !t_idx || !d_idx || !counters || !sums || cap <= 0 || T <= 0 || T > 65535)
        return fail(FPV_ERR_ARG, "%s: null argument, cap <= 0 or T outside 1..65535", who);
    if (e->cfg.precision != FPV_F32) return fail(FPV_ERR_UNSUPPORTED, "%s: fp32 handles", who);
    if (snapshots && snapshot_stride < 2 * e->obs_words) return fail(FPV_ERR_ARG, "%s: snapshot rows hold {mean[O], var[O], ...}", who);
    if (raw && (!rewards || !rew_snapshots || rew_stride < 43)) return fail(FPV_ERR_ARG, "%s: reward scaling needs rewards and snapshot rows of >= 43 words", who);
    FPV_HIP(launch_ledger_scan(e->led_count, e->led_idx, e->led_info, e->led_tag, (const float*)e->led_ret, (const float*)e->led_obs,
                               e->tick, e->tiles, (int)e->led_rows, e->obs_words, snapshots, snapshot_stride, T, clip_obs, eps_norm,
                               tobs, t_idx, d_idx, cap, counters, sums, counters_are_zero != 0, e->merge_ticket + 4, status_out,
                               raw, rewards, rew_snapshots, rew_stride, e->cfg.num_envs, clip_reward, eps_reward, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_ledger_scan(fpv_env* e, const double* snapshots, int64_t snapshot_stride, int32_t T, float clip_obs, float eps_norm,
                    float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                    int32_t counters_are_zero, void* status_out, void* stream) {
    return ledger_scan_impl("fpv_ledger_scan", e, snapshots, snapshot_stride, T, clip_obs, eps_norm, tobs, t_idx, d_idx, cap, counters, sums,
                            counters_are_zero, status_out, nullptr, nullptr, 0, 0.f, 0.f, nullptr, stream);
}

int fpv_ledger_scan_finalize(fpv_env* e, const double* snapshots, int64_t snapshot_stride, int32_t T, float clip_obs, float eps_norm,
                             float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                             int32_t counters_are_zero, void* status_out, const float* raw, const double* reward_snapshots,
                             int64_t reward_snapshot_stride, float clip_reward, float eps_reward, float* rewards, void* stream) {
    if (!raw) return fail(FPV_ERR_ARG, "fpv_ledger_scan_finalize: raw is required (fpv_ledger_scan is the call without the reward scaling)");
    return ledger_scan_impl("fpv_ledger_scan_finalize", e, snapshots, snapshot_stride, T, clip_obs, eps_norm, tobs, t_idx, d_idx, cap,
                            counters, sums, counters_are_zero, status_out, raw, reward_snapshots, reward_snapshot_stride, clip_reward,
                            eps_reward, rewards, stream);
}

int fpv_ledger_scatter(const int32_t* counters, const int32_t* t_idx, const int32_t* d_idx, const float* values, float gamma,
                       float* rewards, int64_t n, int64_t cap, void* stream) {
    if (!counters || !t_idx || !d_idx || !values || !rewards || n <= 0 || cap <= 0)
        return fail(FPV_ERR_ARG, "fpv_ledger_scatter: null argument or non-positive size");
    FPV_HIP(launch_ledger_scatter(counters, t_idx, d_idx, values, gamma, rewards, n, cap, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
            const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* advantages,
            float* returns, double* stats, double* mean_std, void* stream) {
    if (T <= 0 || n <= 0) return fail(FPV_ERR_ARG, "fpv_gae: T and n must be positive");
    if (mean_std && !stats) return fail(FPV_ERR_ARG, "fpv_gae: mean_std needs stats");
    if (!rewards || !values || !episode_starts || !last_values || !last_dones || !advantages || !returns)
        return fail(FPV_ERR_ARG, "fpv_gae: null buffer");
    double* scratch = nullptr;
    if (int rc = scratch_for_current_device(&scratch)) return rc;
    FPV_HIP(launch_gae(T, n, rewards, values, episode_starts, last_values, last_dones, gamma, lam, advantages,
                       returns, stats, mean_std, scratch, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_normalize(int64_t n, int32_t d, const float* x, float* y, double* rms, int32_t update, float clip, float eps,
                  void* stream) {
    if (n <= 0 || d <= 0 || d > 32) return fail(FPV_ERR_ARG, "fpv_normalize: need n > 0 and 0 < d <= 32");
    if (!x || !rms) return fail(FPV_ERR_ARG, "fpv_normalize: null buffer");
    double* scratch = nullptr;
    if (int rc = scratch_for_current_device(&scratch)) return rc;
    FPV_HIP(launch_normalize(n, d, x, y, rms, update, clip, eps, scratch, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_normalize_reward(int64_t n, const float* r, const uint8_t* done, float* ret, float* r_out, double* rms,
                         int32_t update, float gamma, float clip, float eps, void* stream) {
    if (n <= 0) return fail(FPV_ERR_ARG, "fpv_normalize_reward: n must be positive");
    if (!r || !done || !ret || !r_out || !rms) return fail(FPV_ERR_ARG, "fpv_normalize_reward: null buffer");
    double* scratch = nullptr;
    if (int rc = scratch_for_current_device(&scratch)) return rc;
    FPV_HIP(launch_normalize_reward(n, r, done, ret, r_out, rms, update, gamma, clip, eps, scratch,
                                    (hipStream_t)stream));
    return FPV_OK;
}

int fpv_policy_create(int32_t device, fpv_policy** out) {
    if (!out) return fail(FPV_ERR_ARG, "fpv_policy_create: null argument");
    int ndev = 0;
    FPV_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(FPV_ERR_ARG, "fpv_policy_create: device %d not present", device);
    FPV_HIP(hipSetDevice(device));
    fpv_policy* p = new (std::nothrow) fpv_policy();
    if (!p) return fail(FPV_ERR_ALLOC, "fpv_policy_create: host allocation failed");
    p->device = device;
    hipDeviceProp_t prop;
    FPV_HIP(hipGetDeviceProperties(&prop, device));
    p->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p->gemm = 0;
    p->in_dim = 20;
    p->act_dim = 4;
    p->split_image = nullptr;
    p->image16 = nullptr;
    hipError_t e = hipMalloc((void**)&p->image, sizeof(float) * (size_t)policy_image_floats());
    if (e == hipSuccess) e = hipMalloc((void**)&p->split_image, (size_t)policy_split_image_bytes());
#ifdef FPV_EXPERIMENTS
    if (e == hipSuccess) e = hipMalloc((void**)&p->image16, sizeof(float) * (size_t)policy16_image_floats());
#endif
    if (e != hipSuccess) {
        delete p;
        return fail(FPV_ERR_HIP, "fpv_policy_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return FPV_OK;
}

int fpv_policy_destroy(fpv_policy* p) {
    if (!p) return FPV_OK;
    if (p->image) (void)hipFree(p->image);
    if (p->split_image) (void)hipFree(p->split_image);
    if (p->image16) (void)hipFree(p->image16);
    delete p;
    return FPV_OK;
}

int fpv_policy_load(fpv_policy* p, const float* const* weights, void* stream) {
    if (!p || !weights) return fail(FPV_ERR_ARG, "fpv_policy_load: null argument");
    for (int i = 0; i < 13; ++i)
        if (!weights[i]) return fail(FPV_ERR_ARG, "fpv_policy_load: weight pointer %d is null", i);
    FPV_HIP(launch_policy_pack(weights, p->image, p->in_dim, p->act_dim, (hipStream_t)stream));
#ifdef FPV_EXPERIMENTS
    if (p->in_dim <= 20) FPV_HIP(launch_policy16_pack(weights, p->image16, p->in_dim, p->act_dim, (hipStream_t)stream));
#endif
    if ((p->in_dim == 20 && p->act_dim == 4) || (p->in_dim == 12 && p->act_dim == 1))      // the widths the split kernels are built for
        FPV_HIP(launch_policy_split_pack(weights, p->split_image, p->in_dim, p->act_dim, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_policy_set_dims(fpv_policy* p, int32_t in_dim, int32_t act_dim) {
    if (!p) return fail(FPV_ERR_ARG, "fpv_policy_set_dims: null handle");
    if (in_dim < 4 || in_dim > 24 || (in_dim % 4) || act_dim < 1 || act_dim > 4)
        return fail(FPV_ERR_ARG, "fpv_policy_set_dims: need in_dim in {4,8,...,24} and act_dim in 1..4");
    p->in_dim = in_dim;
    p->act_dim = act_dim;
    return FPV_OK;
}

int fpv_policy_set_gemm(fpv_policy* p, int32_t mode) {
    if (!p) return fail(FPV_ERR_ARG, "fpv_policy_set_gemm: null handle");
    if (mode < 0 || mode > 2)
        return fail(FPV_ERR_ARG, "fpv_policy_set_gemm: mode must be 0 (fp32 MFMA), 1 (bf16 6-term split) or 2 (fp32 MFMA, 16-wave kernel)");
#ifndef FPV_EXPERIMENTS
    if (mode == 2) return fail(FPV_ERR_UNSUPPORTED, "fpv_policy_set_gemm: mode 2 (the 16-wave kernel) is part of the FPV_EXPERIMENTS build");
#endif
    p->gemm = mode;
    return FPV_OK;
}

int fpv_policy_forward(fpv_policy* p, const fpv_policy_io* io, void* stream) {
    if (!p || !io) return fail(FPV_ERR_ARG, "fpv_policy_forward: null argument");
    const int nets = io->nets == 0 ? 3 : io->nets;
    if (nets < 1 || nets > 3) return fail(FPV_ERR_ARG, "fpv_policy_forward: nets must be 0..3");
    if (io->n <= 0 || !io->obs || ((nets & 2) && !io->value))
        return fail(FPV_ERR_ARG, "fpv_policy_forward: n, obs and (when V is evaluated) value are required");
    const uintptr_t act_rows = p->act_dim == 4 ? ((uintptr_t)io->action | (uintptr_t)io->clipped | (uintptr_t)io->eps_in |
                                                   (uintptr_t)io->eps_out) : 0;
    if (((uintptr_t)io->obs | (uintptr_t)io->obs_n | act_rows) & 15)
        return fail(FPV_ERR_ARG, "fpv_policy_forward: row buffers must be 16-byte aligned");
    MergeArgs mg{};
    if (io->merge_partials) {
        if (io->merge_rows <= 0 || io->merge_rows > 1024)
            return fail(FPV_ERR_UNSUPPORTED, "fpv_policy_forward: in-launch merge takes 1..1024 partial rows (use fpv_moments_merge beyond)");
        if (p->in_dim > 20 || io->narrow)
            return fail(FPV_ERR_UNSUPPORTED, "fpv_policy_forward: in-launch merge needs in_dim <= 20 and 512-thread workgroups");
        if (!(io->merge_count > 0.0)) return fail(FPV_ERR_ARG, "fpv_policy_forward: merge_count must be positive");
        if ((io->rms && (!io->rms_out || io->rms_out == io->rms)) || (io->ret_rms && (!io->ret_rms_out || io->ret_rms_out == io->ret_rms)))
            return fail(FPV_ERR_ARG, "fpv_policy_forward: merged statistics go to a second buffer (other workgroups still read the first)");
        mg = MergeArgs{io->merge_partials, io->ret_rms, io->rms ? io->rms_out : nullptr, io->ret_rms ? io->ret_rms_out : nullptr,
                       io->merge_snapshot, io->merge_count, (int32_t)io->merge_rows, p->in_dim};
    }
    ScatterArgs sc{};
    if (io->scatter_out) {
        if (!(nets & 2) || !io->scatter_t || !io->scatter_d || io->scatter_stride <= 0)
            return fail(FPV_ERR_ARG, "fpv_policy_forward: scatter needs the V net, scatter_t, scatter_d and a positive scatter_stride");
        sc = ScatterArgs{io->scatter_t, io->scatter_d, io->scatter_out, io->scatter_stride, io->scatter_scale};
    }
    // the bf16-split kernels (built for the two tasks' widths, 20 -> 4 and 12 -> 1) evaluate the nets (both, or one: the rollout tail's V(s_T)) in 512-thread workgroups over
    // a host-side row count; any other request is served by the exact-fp32 kernel (its image is always packed)
    const bool hover_w = p->in_dim == 20 && p->act_dim == 4, yaw_w = p->in_dim == 12 && p->act_dim == 1;
    if (p->gemm == 1 && (hover_w || yaw_w) && !io->narrow && !io->n_dev && !sc.out) {
        FPV_HIP(launch_policy_split_forward(p->split_image, io->obs, io->rms, io->eps_in, io->tick, io->obs_n, io->action,
                                            io->clipped, io->value, io->logp, io->eps_out, io->n, io->seed, io->id_offset,
                                            io->clip_obs, io->eps_norm, io->deterministic, p->cus, nets, yaw_w ? 1 : 0, mg, (hipStream_t)stream));
        return FPV_OK;
    }
#ifdef FPV_EXPERIMENTS
    if (p->gemm == 2 && nets == 3 && !io->narrow && !mg.rows && !io->eps_in && !io->deterministic && !io->n_dev && !sc.out && p->in_dim <= 20) {
        // the rollout's launch shape (both nets, in-kernel sampling): 16-wave workgroups on 16x16x4 tiles
        FPV_HIP(launch_policy16_forward(p->image16, io->obs, io->rms, io->tick, io->obs_n, io->action, io->clipped, io->value,
                                        io->logp, io->eps_out, io->n, io->seed, io->id_offset, io->clip_obs, io->eps_norm,
                                        p->cus, p->in_dim, p->act_dim, (hipStream_t)stream));
        return FPV_OK;
    }
#else
    if (io->narrow) return fail(FPV_ERR_UNSUPPORTED, "fpv_policy_forward: narrow (256-thread) launches are part of the FPV_EXPERIMENTS build");
#endif
    FPV_HIP(launch_policy_forward(p->image, io->obs, io->rms, io->eps_in, io->tick, io->obs_n, io->action, io->clipped,
                                  io->value, io->logp, io->eps_out, io->n, io->seed, io->id_offset, io->clip_obs,
                                  io->eps_norm, io->deterministic, p->cus, nets, io->narrow ? 1 : 0, p->in_dim, p->act_dim, mg,
                                  io->n_dev, sc, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_rollout_step(fpv_env* e, fpv_policy* p, const fpv_policy_io* io, float* reward, uint8_t* terminated, uint8_t* truncated,
                     const fpv_step_extras* ex, void* stream) {
    if (!e || !p || !io) return fail(FPV_ERR_ARG, "fpv_rollout_step: null argument");
    if (!io->obs || (!io->value && io->nets != 1) || !reward || !terminated || !truncated) return fail(FPV_ERR_ARG, "fpv_rollout_step: null buffer");
    if (e->cfg.precision != FPV_F32 || e->cfg.noise_ratio == 0.0)
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: fp32 handles with motor noise only");
    if (e->cfg.num_envs % 256 != 0) return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: num_envs must be a multiple of 256");
    if (io->n != e->cfg.num_envs) return fail(FPV_ERR_ARG, "fpv_rollout_step: io->n must equal num_envs");
    const bool yaw = e->cfg.task == FPV_TASK_YAW;
    if (p->in_dim != e->obs_words || p->act_dim != e->act_words || (p->gemm != 0 && p->gemm != 1))
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: policy widths must be the task's (%d / %d), gemm mode 0 or 1 (the bf16 split)",
                    e->obs_words, e->act_words);
    if (io->eps_in || io->eps_out || io->deterministic || io->narrow || io->n_dev || io->scatter_out ||
        (io->nets != 0 && io->nets != 3 && !(io->nets == 1 && p->gemm == 1)))
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: eps_in / eps_out / deterministic / narrow / n_dev / scatter / V alone are not served; "
                                         "pi alone (nets = 1, the critic deferred) in gemm mode 1 only");
    if (ex && (ex->obs_rms || ex->ret_rms || ex->snapshot))
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: the statistics update rides in the next call's io->merge_* fields");
    if (((uintptr_t)io->obs | (uintptr_t)io->obs_n | (yaw ? 0 : ((uintptr_t)io->action | (uintptr_t)io->clipped))) & 15)
        return fail(FPV_ERR_ARG, "fpv_rollout_step: row buffers must be 16-byte aligned");
    MergeArgs mg{};
    if (io->merge_partials) {
        if (io->merge_rows <= 0 || io->merge_rows > 1024)
            return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_step: in-launch merge takes 1..1024 partial rows");
        if (!(io->merge_count > 0.0)) return fail(FPV_ERR_ARG, "fpv_rollout_step: merge_count must be positive");
        if ((io->rms && (!io->rms_out || io->rms_out == io->rms)) || (io->ret_rms && (!io->ret_rms_out || io->ret_rms_out == io->ret_rms)))
            return fail(FPV_ERR_ARG, "fpv_rollout_step: merged statistics go to a second buffer (other workgroups still read the first)");
        if (ex && ex->partials == io->merge_partials)
            return fail(FPV_ERR_ARG, "fpv_rollout_step: extras->partials must not be the table this call merges (alternate two tables)");
        mg = MergeArgs{io->merge_partials, io->ret_rms, io->rms ? io->rms_out : nullptr, io->ret_rms ? io->ret_rms_out : nullptr,
                       io->merge_snapshot, io->merge_count, (int32_t)io->merge_rows, p->in_dim};
    }
    StepArgs<float> a{};
    a.state = (Vec4<float>*)e->state;
    a.actions = nullptr; a.obs = const_cast<float*>(io->obs); a.reward = reward; a.terminated = terminated; a.truncated = truncated;
    a.tick = e->tick;
    a.tmpl = (const float*)e->tmpl;
    a.led_count = e->led_count; a.led_idx = e->led_idx; a.led_info = e->led_info; a.led_tag = e->led_tag;
    a.led_ret = (float*)e->led_ret; a.led_obs = (float*)e->led_obs; a.led_rows = e->led_rows;
    a.n = e->cfg.num_envs;
    a.seed = e->cfg.seed;
    a.id_offset = e->cfg.id_offset;
    a.sigma = (float)e->cfg.noise_ratio;
    a.floor_check = e->cfg.floor_check;
    a.domain_rand = e->cfg.domain_rand;
    a.detector = e->cfg.detector == FPV_DETECTOR_PIXEL ? 1 : 0;   // kernel-side switch: non-zero = pixel rules
    a.done = ex ? ex->done : nullptr;
    a.ret = ex ? ex->ret : nullptr;
    a.partials = ex ? ex->partials : nullptr;
    a.gamma = ex ? ex->gamma : 0.0f;
    a.mg_count = (double)e->cfg.num_envs;
    if (p->gemm == 1)
        FPV_HIP(launch_rollout_step_split(p->split_image, io->rms, io->obs_n, io->action, io->clipped, io->value, io->logp, io->seed,
                                          io->id_offset, io->clip_obs, io->eps_norm, mg, a, yaw ? 1 : 0, io->nets == 1 ? 0 : 1, e->stamps, (hipStream_t)stream));
    else
        FPV_HIP(launch_rollout_step(p->image, io->rms, io->obs_n, io->action, io->clipped, io->value, io->logp, io->seed, io->id_offset,
                                    io->clip_obs, io->eps_norm, mg, a, yaw ? 1 : 0, e->stamps, (hipStream_t)stream));
    return FPV_OK;
}

#ifdef FPV_EXPERIMENTS   // the persistent launch: measured slower than one launch per step
int fpv_rollout_steps(fpv_env* e, fpv_policy* p, const fpv_rollout_io* io, void* stream) {
    if (!e || !p || !io) return fail(FPV_ERR_ARG, "fpv_rollout_steps: null argument");
    if (io->T <= 0) return fail(FPV_ERR_ARG, "fpv_rollout_steps: T must be positive");
    if (!io->obs || !io->value || !io->reward || !io->terminated || !io->truncated || !io->status)
        return fail(FPV_ERR_ARG, "fpv_rollout_steps: obs, value, reward, terminated, truncated and status are required");
    if (e->cfg.precision != FPV_F32 || e->cfg.task != FPV_TASK_HOVER || e->cfg.noise_ratio == 0.0)
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_steps: fp32 hover handles with motor noise only");
    if (e->cfg.detector == FPV_DETECTOR_PIXEL)
        // (the pixel detector's work-item exchange lives above the part of the weight image this kernel stages again per step)
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_steps: the persistent launch serves FPV_DETECTOR_ANALYTIC handles only");
    if (e->cfg.num_envs % 256 != 0 || e->cfg.num_envs / 256 > p->cus)
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_steps: num_envs must be a multiple of 256 and at most 256 x %d (one workgroup per CU)", p->cus);
    if (p->in_dim != 20 || p->act_dim != 4 || p->gemm != 0)
        return fail(FPV_ERR_UNSUPPORTED, "fpv_rollout_steps: policy widths 20 / 4 in gemm mode 0 only");
    if (((uintptr_t)io->obs | (uintptr_t)io->obs_n | (uintptr_t)io->action) & 15)
        return fail(FPV_ERR_ARG, "fpv_rollout_steps: row buffers must be 16-byte aligned");
    const bool norm = io->rms || io->ret_rms;
    if (!io->partials[0] || !io->partials[1] || io->partials[0] == io->partials[1])
        return fail(FPV_ERR_ARG, "fpv_rollout_steps: two different partials tables are required");
    if (io->T > 1 && norm) {
        if ((io->rms && (!io->rms_out[0] || !io->rms_out[1])) || (io->ret_rms && (!io->ret_rms_out[0] || !io->ret_rms_out[1])))
            return fail(FPV_ERR_ARG, "fpv_rollout_steps: rms_out / ret_rms_out are required with their statistics");
        if (io->snapshots && io->snapshot_stride < FPV_PARTIAL_WORDS)
            return fail(FPV_ERR_ARG, "fpv_rollout_steps: snapshot_stride must be at least FPV_PARTIAL_WORDS");
    }
    StepArgs<float> a{};
    a.state = (Vec4<float>*)e->state;
    a.obs = io->obs; a.terminated = io->terminated; a.truncated = io->truncated;
    a.tick = e->tick;
    a.tmpl = (const float*)e->tmpl;
    a.led_count = e->led_count; a.led_idx = e->led_idx; a.led_info = e->led_info; a.led_tag = e->led_tag;
    a.led_ret = (float*)e->led_ret; a.led_obs = (float*)e->led_obs; a.led_rows = e->led_rows;
    a.n = e->cfg.num_envs;
    a.seed = e->cfg.seed;
    a.id_offset = e->cfg.id_offset;
    a.sigma = (float)e->cfg.noise_ratio;
    a.floor_check = e->cfg.floor_check;
    a.domain_rand = e->cfg.domain_rand;
    a.detector = e->cfg.detector == FPV_DETECTOR_PIXEL ? 1 : 0;   // kernel-side switch: non-zero = pixel rules
    a.ret = io->ret;
    a.gamma = io->gamma;
    a.mg_count = (double)e->cfg.num_envs;
    RollPersistHost h{};
    h.image = p->image; h.pol_seed = io->seed; h.pol_id_offset = io->id_offset; h.clip_obs = io->clip_obs; h.eps_norm = io->eps_norm;
    h.T = io->T;
    h.obs_n = io->obs_n; h.action = io->action; h.value = io->value; h.logp = io->logp; h.reward = io->reward; h.done = io->done;
    h.rms0 = io->rms; h.ret_rms0 = io->ret_rms;
    h.rms_out[0] = io->rms_out[0]; h.rms_out[1] = io->rms_out[1];
    h.ret_rms_out[0] = io->ret_rms_out[0]; h.ret_rms_out[1] = io->ret_rms_out[1];
    h.snapshots = io->snapshots; h.snap_stride = io->snapshot_stride;
    h.rows[0] = io->partials[0]; h.rows[1] = io->partials[1];
    h.sync = e->rollout_sync; h.status = io->status;
    // 2 s of the 100 MHz counter: far beyond any step, short enough to end a stuck launch (fpv_debug_rollout_timeout overrides)
    h.timeout_ticks = e->timeout_ticks ? e->timeout_ticks : 200000000ull;
    h.stamps = e->stamps;
    FPV_HIP(launch_rollout_persistent(h, a, (hipStream_t)stream));
    return FPV_OK;
}
#endif   // FPV_EXPERIMENTS

int fpv_ppo_workspace_bytes(fpv_policy* p, int64_t rows, int64_t* bytes) {
    if (!p || !bytes || rows <= 0) return fail(FPV_ERR_ARG, "fpv_ppo_workspace_bytes: null argument or rows <= 0");
    *bytes = ppo_workspace_floats(rows, p->cus) * (int64_t)sizeof(float);
    return FPV_OK;
}

int fpv_ppo_grad(fpv_policy* p, const fpv_ppo_io* io, void* stream) {
    if (!p || !io) return fail(FPV_ERR_ARG, "fpv_ppo_grad: null argument");
    if (io->rows <= 0 || !io->obs || !io->actions || !io->old_logp || !io->advantages || !io->returns || !io->stats ||
        !io->workspace)
        return fail(FPV_ERR_ARG, "fpv_ppo_grad: rows, obs, actions, old_logp, advantages, returns, stats and workspace are required");
    for (int i = 0; i < 13; ++i)
        if (!io->weights[i] || !io->grads[i]) return fail(FPV_ERR_ARG, "fpv_ppo_grad: weight / gradient pointer %d is null", i);
    const uintptr_t act_rows = p->act_dim == 4 ? (uintptr_t)io->actions : 0;
    if (((uintptr_t)io->obs | act_rows | (uintptr_t)io->workspace) & 15)
        return fail(FPV_ERR_ARG, "fpv_ppo_grad: row buffers and workspace must be 16-byte aligned");
    if (io->storage != 0 && io->storage != 1) return fail(FPV_ERR_ARG, "fpv_ppo_grad: storage must be 0 (fp32) or 1 (bf16)");
    if (io->path < FPV_PPO_ONE_KERNEL || io->path > FPV_PPO_TWO_KERNELS_BF16X3)
        return fail(FPV_ERR_ARG, "fpv_ppo_grad: path must be FPV_PPO_ONE_KERNEL, FPV_PPO_TWO_KERNELS, FPV_PPO_TWO_KERNELS_BF16X6 or FPV_PPO_TWO_KERNELS_BF16X3");
    if (io->path != FPV_PPO_TWO_KERNELS && io->storage == 1)
        return fail(FPV_ERR_ARG, "fpv_ppo_grad: bf16 storage exists on the FPV_PPO_TWO_KERNELS path only (nothing is stored on the "
                                 "one-kernel path; the bf16 x 6 products split fp32 operands)");
    const int64_t need = ppo_workspace_floats(io->rows, p->cus) * (int64_t)sizeof(float);
    if (io->workspace_bytes < need)
        return fail(FPV_ERR_ARG, "fpv_ppo_grad: workspace of %lld bytes, %lld needed (fpv_ppo_workspace_bytes)",
                    (long long)io->workspace_bytes, (long long)need);
    FPV_HIP(launch_ppo_grad(io->weights, io->grads, io->index, io->obs, io->actions, io->old_logp, io->advantages, io->returns,
                            io->rows, io->clip_range, io->vf_coef, io->ent_coef, io->stats, (float*)io->workspace, p->in_dim,
                            p->act_dim, p->cus, io->storage, io->path, (hipStream_t)stream));
    return FPV_OK;
}

static int adam_clip_step(const char* who, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                          float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                          const int32_t* gate, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n <= 0)
        return fail(FPV_ERR_ARG, "%s: null argument or n <= 0", who);
    if (n > (1 << 20))
        return fail(FPV_ERR_UNSUPPORTED, "%s: built for policy-sized tensors (n <= 2^20): every workgroup recomputes the norm", who);
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps > 0.f))
        return fail(FPV_ERR_ARG, "%s: need lr >= 0, betas in [0, 1), eps > 0", who);
    FPV_HIP(launch_adam_clip(param, grad, exp_avg, exp_avg_sq, step, grad_norm, n, lr, beta1, beta2, eps, max_grad_norm, gate,
                             (hipStream_t)stream));
    return FPV_OK;
}

int fpv_adam_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                       float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                       void* stream) {
    return adam_clip_step("fpv_adam_clip_step", param, grad, exp_avg, exp_avg_sq, step, grad_norm, n, lr, beta1, beta2, eps,
                          max_grad_norm, nullptr, stream);
}

int fpv_adam_clip_step_gated(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t* step,
                             float* grad_norm, int64_t n, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                             const int32_t* gate, void* stream) {
    return adam_clip_step("fpv_adam_clip_step_gated", param, grad, exp_avg, exp_avg_sq, step, grad_norm, n, lr, beta1, beta2, eps,
                          max_grad_norm, gate, stream);
}

int fpv_ppo_kl_gate(const float* epoch_stats, int32_t n_minibatches, float limit, int32_t* gate, void* stream) {
    if (!epoch_stats || !gate || n_minibatches <= 0) return fail(FPV_ERR_ARG, "fpv_ppo_kl_gate: null argument or n_minibatches <= 0");
    FPV_HIP(launch_ppo_kl_gate(epoch_stats, n_minibatches, limit, gate, (hipStream_t)stream));
    return FPV_OK;
}

int fpv_shuffle_indices(int64_t* out, int64_t n, uint64_t seed, uint64_t counter, void* stream) {
    if (!out || n <= 0) return fail(FPV_ERR_ARG, "fpv_shuffle_indices: null argument or n <= 0");
    if (n > ((int64_t)1 << 40)) return fail(FPV_ERR_UNSUPPORTED, "fpv_shuffle_indices: n <= 2^40");
    FPV_HIP(launch_shuffle_indices(out, n, seed, counter, (hipStream_t)stream));
    return FPV_OK;
}

}  // extern "C"
#pragma GCC visibility pop
