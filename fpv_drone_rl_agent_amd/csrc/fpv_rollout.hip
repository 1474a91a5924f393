// fpv_rollout.hip — PPO rollout primitives on device (gfx950): GAE backward scan with the
// advantage statistics of the one all-gather per update, and the VecNormalize running-moment
// normaliser.  All of these are HBM-streaming kernels: coalesced time-major / row-major
// accesses, fp64 accumulators, and two-stage reductions whose order is fixed so that the
// results are bitwise reproducible run to run (no float atomics).
//
// Semantics: SB3 RolloutBuffer.compute_returns_and_advantage and VecNormalize /
// RunningMeanStd as used by simulation/train_hover.py:42-60.
#include "fpv_internal.h"
#include <type_traits>

namespace fpv {

constexpr int kRedBlock = 256;
constexpr int kMaxRedBlocks = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of two doubles in a fixed order; result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b, double* lds /* [2 * waves] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { lds[2 * wave] = a; lds[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0; b = 0.0;
        for (int w = 0; w < waves; ++w) { a += lds[2 * w]; b += lds[2 * w + 1]; }
    }
}

// ---------------------------------------------------------------- K3 GAE
// One lane per env; every load/store at fixed t is a contiguous run over envs.
__global__ __launch_bounds__(kRedBlock) void gae_kernel(int64_t T, int64_t n, const float* __restrict__ rewards,
                                                        const float* __restrict__ values,
                                                        const uint8_t* __restrict__ starts,
                                                        const float* __restrict__ last_values,
                                                        const uint8_t* __restrict__ last_dones, float gamma, float lam,
                                                        float* __restrict__ adv, float* __restrict__ returns,
                                                        double* __restrict__ partial, double* __restrict__ stats,
                                                        double* __restrict__ mean_std, double count, uint32_t* __restrict__ ticket) {
    __shared__ double lds[2 * (kRedBlock / 64) + 130];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float next_v = last_values[i];
        float next_nt = 1.0f - (float)last_dones[i];
        float last = 0.0f;
        // The scan is a dependent chain, its inputs are not.  The rows of 16 time steps are requested as a block, and the next
        // block's rows before the current block is worked through, so that a rollout of <= 32 steps waits out ONE memory round
        // trip (a plain loop waits one per step: ~15 us for 32 steps at 65 536 drones, one wave per SIMD).  Same operations in
        // the same order as the plain loop.
        constexpr int Q = 16;
        float vA[Q], rA[Q], vB[Q], rB[Q];
        uint8_t sA[Q], sB[Q];
        // (requests are unconditional — rows below 0 are clamped to row 0 and not used — so that no load sits under a branch:
        // the compiler waits at every conditional load)
        auto request = [&](float (&v_)[Q], float (&r_)[Q], uint8_t (&s_)[Q], int64_t t) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int64_t e = (t - q > 0 ? t - q : 0) * n + i;
                v_[q] = values[e]; r_[q] = rewards[e]; s_[q] = starts[e];
            }
        };
        auto step = [&](float v, float r, uint8_t st, int64_t e) {
            const float delta = r + gamma * next_v * next_nt - v;
            last = delta + gamma * lam * next_nt * last;
            adv[e] = last;
            returns[e] = last + v;
            s1 += (double)last;
            s2 += (double)last * (double)last;
            next_v = v;
            next_nt = 1.0f - (float)st;
        };
        auto work = [&](const float (&v_)[Q], const float (&r_)[Q], const uint8_t (&s_)[Q], int64_t t, int cnt) {
            if (cnt == Q) {
#pragma unroll
                for (int q = 0; q < Q; ++q) step(v_[q], r_[q], s_[q], (t - q) * n + i);
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (q < cnt) step(v_[q], r_[q], s_[q], (t - q) * n + i);
            }
        };
        int64_t t = T - 1;
        int cnt = t + 1 < Q ? (int)(t + 1) : Q;
        request(vA, rA, sA, t);
        while (cnt > 0) {
            int64_t tn = t - cnt;
            int cn = tn + 1 < Q ? (int)(tn + 1) : Q;
            if (cn > 0) request(vB, rB, sB, tn);
            work(vA, rA, sA, t, cnt);
            if (cn <= 0) break;
            t = tn - cn;
            cnt = t + 1 < Q ? (int)(t + 1) : Q;
            if (cnt > 0) request(vA, rA, sA, t);
            work(vB, rB, sB, tn, cn);
        }
    }
    if (partial) {
        // The block sums leave write-through, the block takes a ticket, and the block that draws the last one adds all of them
        // in a fixed order — 64 strided sums, then those in turn: the same bits whichever block it is — so the reduction needs no
        // launch of its own.
        block_sum2(s1, s2, lds);
        int* flag = reinterpret_cast<int*>(lds + 2 * (kRedBlock / 64));
        if (threadIdx.x == 0) {
            row_store_sc1(partial + 2 * blockIdx.x, s1);
            row_store_sc1(partial + 2 * blockIdx.x + 1, s2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = t == gridDim.x - 1u;
        }
        __syncthreads();
        if (*flag == 0) return;                          // uniform
        double* l1 = lds + 2 * (kRedBlock / 64) + 1, *l2 = l1 + 64;   // (word 0 of this area is the flag)
        __syncthreads();
        if (threadIdx.x < 64) {
            double a1 = 0.0, a2 = 0.0;
            for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) { a1 += row_load_sc1(partial + 2 * b); a2 += row_load_sc1(partial + 2 * b + 1); }
            l1[threadIdx.x] = a1;
            l2[threadIdx.x] = a2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a1 = 0.0, a2 = 0.0;
            for (int w = 0; w < 64; ++w) { a1 += l1[w]; a2 += l2[w]; }
            stats[0] = count; stats[1] = a1; stats[2] = a2;
            if (mean_std) {                              // one rank: advantage_stats_kernel's arithmetic on this shard's sums
                const double mean = a1 / count;
                double var = a2 / count - mean * mean;
                if (var < 0.0) var = 0.0;
                mean_std[0] = mean;
                mean_std[1] = sqrt(var);
            }
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
        }
    }
}

// {count, sum A, sum A^2} per rank, added in rank order -> {mean, std} (population std, as SB3 normalises advantages)
__global__ void advantage_stats_kernel(const double* __restrict__ gathered, int world, int stride, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double c = 0.0, s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < world; ++r) { c += gathered[r * stride]; s1 += gathered[r * stride + 1]; s2 += gathered[r * stride + 2]; }
    const double mean = s1 / c;
    double var = s2 / c - mean * mean;
    if (var < 0.0) var = 0.0;
    out[0] = mean;
    out[1] = sqrt(var);
}

// Several ranks: VecNormalize's statistics in the same payload (SURVEY §8e "can ride in the same payload").  Every rank starts a
// rollout from the same statistics `base` = {mean[d], var[d], count} and folds its own drones' samples into them step by step; its
// row of the gathered payload carries what it holds afterwards.  What rank r added is recovered from (base, after_r) —
//     c_b = c_1 - c_0,  m_b = (m_1 c_1 - m_0 c_0) / c_b,  M2_b = v_1 c_1 - v_0 c_0 - (m_b - m_0)^2 c_0 c_b / c_1
// (the inverse of RunningMeanStd.update_from_moments) — and the ranks' batches are merged into `base` in rank order with that
// very update: every rank derives the same bits, and they are the moments of the union of all ranks' samples.
// One thread per channel; gathered rows: [adv stats 3][obs mean d, var d, count][ret mean, var, count].
// The bases may BE the outputs (in-place use): the four pointers are not restrict-qualified, every thread reads its base words
// before any thread of the launch — one wave — stores.
__global__ void rank_merge_kernel(const double* __restrict__ gathered, int world, int stride, int d, const double* base_obs,
                                  const double* base_ret, double* out_obs, double* out_ret) {
    const int c = threadIdx.x;
    const bool obs_ch = c < d && base_obs != nullptr, ret_ch = c == d && base_ret != nullptr;
    if (!obs_ch && !ret_ch) return;
    // (branches, not selects of two loads: either statistics may be left out.  Round 3 saw one fault at address 0 in this kernel while
    // it was being changed — start() passes base_ret = null; the select form was the suspect, no ISA of the faulting build was
    // kept, so the cause is unproven.  tests/test_rollout_ops_gpu.py calls the obs-only, ret-only and null-base forms directly.)
    double m0, v0, c0;
    if (obs_ch) { m0 = base_obs[c]; v0 = base_obs[d + c]; c0 = base_obs[2 * d]; }
    else { m0 = base_ret[0]; v0 = base_ret[1]; c0 = base_ret[2]; }
    double mean = m0, var = v0, count = c0;
    const int ic = obs_ch ? c : 0, iv = obs_ch ? d + c : 1, in = obs_ch ? 2 * d : 2;       // this thread's words of a gathered row
    for (int r = 0; r < world; ++r) {
        const double* row = gathered + (int64_t)r * stride + 3 + (obs_ch ? 0 : 2 * d + 1);
        const double m1 = row[ic], v1 = row[iv], c1 = row[in];
        const double cb = c1 - c0;
        if (!(cb > 0.0)) continue;                       // this rank added nothing
        const double mb = (m1 * c1 - m0 * c0) / cb, db = mb - m0;
        double vb = (v1 * c1 - v0 * c0 - db * db * c0 * cb / c1) / cb;
        if (vb < 0.0) vb = 0.0;
        const double delta = mb - mean, tot = count + cb;                                       // RunningMeanStd.update_from_moments
        const double m2 = var * count + vb * cb + delta * delta * count * cb / tot;
        mean += delta * cb / tot;
        var = m2 / tot;
        count = tot;
    }
    if (obs_ch) {
        out_obs[c] = mean; out_obs[d + c] = var;
        if (c == 0) out_obs[2 * d] = count;
    } else {
        out_ret[0] = mean; out_ret[1] = var; out_ret[2] = count;
    }
}

hipError_t launch_rank_merge(const double* gathered, int world, int stride, int d, const double* base_obs, const double* base_ret,
                             double* out_obs, double* out_ret, hipStream_t s) {
    hipLaunchKernelGGL(rank_merge_kernel, dim3(1), dim3(64), 0, s, gathered, world, stride, d, base_obs, base_ret, out_obs, out_ret);
    return hipGetLastError();
}

hipError_t launch_advantage_stats(const double* gathered, int world, int stride, double* out, hipStream_t s) {
    hipLaunchKernelGGL(advantage_stats_kernel, dim3(1), dim3(64), 0, s, gathered, world, stride, out);
    return hipGetLastError();
}

hipError_t launch_gae(int64_t T, int64_t n, const float* rewards, const float* values, const uint8_t* episode_starts,
                      const float* last_values, const uint8_t* last_dones, float gamma, float lam, float* adv,
                      float* returns, double* stats, double* mean_std, double* scratch, hipStream_t s) {
    int64_t blocks = (n + kRedBlock - 1) / kRedBlock;
    if (blocks > kMaxRedBlocks * 2) blocks = kMaxRedBlocks * 2;
    // the ticket of the in-launch reduction: the last word of the device's reduction scratch (zeroed at allocation, reset by
    // the block that finishes the reduction)
    uint32_t* ticket = reinterpret_cast<uint32_t*>(scratch + kScratchDoubles - 1);
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)blocks), dim3(kRedBlock), 0, s, T, n, rewards, values,
                       episode_starts, last_values, last_dones, gamma, lam, adv, returns, stats ? scratch : nullptr, stats,
                       stats ? mean_std : nullptr, (double)T * (double)n, ticket);
    return hipGetLastError();
}

// ---------------------------------------------------------------- K4 running moments (obs)
// x is [n][d] row-major read as a flat stream; blockDim = lcm(64, d) so that a thread's
// channel (flat index mod d) is fixed while it strides through the stream.
__global__ void moments_partial_kernel(int64_t total, int d, const float* __restrict__ x,
                                       double* __restrict__ partial) {
    extern __shared__ double lds[];  // [2 * blockDim]
    double s1 = 0.0, s2 = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += stride) {
        const double v = (double)x[f];
        s1 += v;
        s2 += v * v;
    }
    lds[threadIdx.x] = s1;
    lds[blockDim.x + threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < d) {
        double a = 0.0, b = 0.0;
        for (int t = threadIdx.x; t < (int)blockDim.x; t += d) { a += lds[t]; b += lds[blockDim.x + t]; }
        partial[(int64_t)blockIdx.x * 2 * d + threadIdx.x] = a;
        partial[(int64_t)blockIdx.x * 2 * d + d + threadIdx.x] = b;
    }
}

// SB3 RunningMeanStd.update_from_moments (parallel-variance merge).  kMergeWays threads per
// channel walk the partial table in a fixed interleave, then one of them folds the kMergeWays
// sums in order: deterministic, and ~blocks / kMergeWays dependent loads deep instead of blocks.
constexpr int kMergeWays = 16;
__global__ __launch_bounds__(32 * kMergeWays) void moments_merge_kernel(const double* __restrict__ partial, int blocks,
                                                                        int d, double batch_count,
                                                                        double* rms /* mean[d], var[d], count */) {
    __shared__ double l1[32 * kMergeWays], l2[32 * kMergeWays];
    const int c = threadIdx.x % 32, j = threadIdx.x / 32;
    const double count = rms[2 * d];
    double s1 = 0.0, s2 = 0.0;
    if (c < d)
        for (int b = j; b < blocks; b += kMergeWays) {
            s1 += partial[(int64_t)b * 2 * d + c];
            s2 += partial[(int64_t)b * 2 * d + d + c];
        }
    l1[threadIdx.x] = s1;
    l2[threadIdx.x] = s2;
    __syncthreads();
    if (j == 0 && c < d) {
        s1 = 0.0; s2 = 0.0;
        for (int w = 0; w < kMergeWays; ++w) { s1 += l1[w * 32 + c]; s2 += l2[w * 32 + c]; }
        const double bm = s1 / batch_count;
        double bv = s2 / batch_count - bm * bm;
        if (bv < 0.0) bv = 0.0;
        const double mean = rms[c], var = rms[d + c];
        const double delta = bm - mean, tot = count + batch_count;
        rms[c] = mean + delta * batch_count / tot;
        rms[d + c] = (var * count + bv * batch_count + delta * delta * count * batch_count / tot) / tot;
    }
    if (threadIdx.x == 0) rms[2 * d] = count + batch_count;
}

__global__ void normalize_apply_kernel(int64_t total, int d, const float* __restrict__ x, float* __restrict__ y,
                                       const double* __restrict__ rms, float clip, float eps) {
    __shared__ float mu[32], inv[32];
    if ((int)threadIdx.x < d) {
        mu[threadIdx.x] = (float)rms[threadIdx.x];
        inv[threadIdx.x] = (float)(1.0 / sqrt(rms[d + threadIdx.x] + (double)eps));
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += stride) {
        const int c = (int)(f % d);
        y[f] = fminf(fmaxf((x[f] - mu[c]) * inv[c], -clip), clip);
    }
}

static int lcm64(int d) {
    int a = 64, b = d;
    while (b) { int t = a % b; a = b; b = t; }
    return 64 / a * d;
}

hipError_t launch_normalize(int64_t n, int32_t d, const float* x, float* y, double* rms, int32_t update, float clip,
                            float eps, double* scratch, hipStream_t s) {
    const int64_t total = n * d;
    const int block = lcm64(d);
    if (update) {
        int64_t blocks = (total + (int64_t)block * 8 - 1) / ((int64_t)block * 8);
        if (blocks > kMaxRedBlocks) blocks = kMaxRedBlocks;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(moments_partial_kernel, dim3((unsigned)blocks), dim3(block), 2 * block * sizeof(double), s,
                           total, (int)d, x, scratch);
        hipLaunchKernelGGL(moments_merge_kernel, dim3(1), dim3(32 * kMergeWays), 0, s, scratch, (int)blocks, (int)d,
                           (double)n, rms);
    }
    if (y) {
        int64_t blocks = (total + 256 * 4 - 1) / (256 * 4);
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(normalize_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, total, (int)d, x, y, rms,
                           clip, eps);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- VecNormalize reward path
__global__ __launch_bounds__(kRedBlock) void return_update_kernel(int64_t n, const float* __restrict__ r,
                                                                  float* __restrict__ ret, float gamma,
                                                                  double* __restrict__ partial) {
    __shared__ double lds[2 * (kRedBlock / 64)];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = ret[i] * gamma + r[i];
        ret[i] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
    }
    block_sum2(s1, s2, lds);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s1; partial[2 * blockIdx.x + 1] = s2; }
}

__global__ __launch_bounds__(kRedBlock) void reward_apply_kernel(int64_t n, const float* __restrict__ r,
                                                                 const uint8_t* __restrict__ done,
                                                                 float* __restrict__ ret, float* __restrict__ r_out,
                                                                 const double* __restrict__ rms, float clip, float eps) {
    const float inv = (float)(1.0 / sqrt(rms[1] + (double)eps));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        r_out[i] = fminf(fmaxf(r[i] * inv, -clip), clip);
        if (done[i]) ret[i] = 0.0f;
    }
}

// ---------------------------------------------------------------- fused-step moment merge
// Stage 1 (only when there are many tiles): G workgroups each fold a contiguous range of tile rows.
__global__ __launch_bounds__(256) void tile_partials_reduce_kernel(const double* __restrict__ partials, int64_t tiles,
                                                                   int64_t per_group, double* __restrict__ group, int P) {
    __shared__ double l[4][kPartialWords];
    const int c = threadIdx.x % 64, j = threadIdx.x / 64;   // 4 ways x 64 lanes (P = 2 O + 4 <= 44 used)
    const int64_t t0 = (int64_t)blockIdx.x * per_group, t1 = t0 + per_group < tiles ? t0 + per_group : tiles;
    double acc = 0.0;
    if (c < P)
        for (int64_t t = t0 + j; t < t1; t += 4) acc += partials[t * P + c];
    if (c < P) l[j][c] = acc;
    __syncthreads();
    if (j == 0 && c < P) group[(int64_t)blockIdx.x * P + c] = (l[0][c] + l[1][c]) + (l[2][c] + l[3][c]);
}

// Stage 2: one workgroup merges the rows (P = 2 O + 4 words: {sum obs[O], sum obs^2[O], sum ret, sum ret^2, 0, 0}) into
// the running statistics of SB3's RunningMeanStd (obs: O channels, {mean[O], var[O], count}; ret: {mean, var, count}) and
// snapshots {obs statistics [2 O + 1] from word 0, ret statistics at words 41..43}.
// 23 ways x P channels (1 012 of the 1 024 threads for the hover task's P = 44); consecutive threads read consecutive
// doubles, and each thread keeps its loads independent so that the whole table costs one memory round trip.
constexpr int kTileMergeWays = 23;
__global__ __launch_bounds__(1024) void tile_partials_merge_kernel(const double* __restrict__ rows, int nrows,
                                                                   double batch_count, const double* obs_rms_in,
                                                                   const double* ret_rms_in, double* obs_rms,
                                                                   double* ret_rms, double* snapshot, int O) {
    __shared__ double l[kTileMergeWays][64];
    const int P = 2 * O + 4;
    const int j = threadIdx.x / P, c = threadIdx.x % P;
    if (j < kTileMergeWays) {
        // nrows <= 1024 (more tiles go through stage 1): every load of this thread is requested before the first
        // add, so the whole table costs one memory round trip; the adds then run in row order
        constexpr int kIter = (1024 + kTileMergeWays - 1) / kTileMergeWays;   // 45
        double v[kIter];
#pragma unroll
        for (int q = 0; q < kIter; ++q) {
            const int b = j + q * kTileMergeWays;
            v[q] = b < nrows ? rows[(int64_t)b * P + c] : 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < kIter; ++q) acc += v[q];
        l[j][c] = acc;
    }
    const double obs_count = obs_rms ? obs_rms_in[2 * O] : 0.0, ret_count = ret_rms ? ret_rms_in[2] : 0.0;
    __syncthreads();
    if (j == 0) {
        double acc = 0.0;
        for (int w = 0; w < kTileMergeWays; ++w) acc += l[w][c];
        l[0][c] = acc;
    }
    __syncthreads();
    if (j != 0) return;
    if (c < O && obs_rms) {
        const double bm = l[0][c] / batch_count;
        double bv = l[0][O + c] / batch_count - bm * bm;
        if (bv < 0.0) bv = 0.0;
        const double mean = obs_rms_in[c], var = obs_rms_in[O + c], delta = bm - mean, tot = obs_count + batch_count;
        const double nm = mean + delta * batch_count / tot;
        const double nv = (var * obs_count + bv * batch_count + delta * delta * obs_count * batch_count / tot) / tot;
        obs_rms[c] = nm; obs_rms[O + c] = nv;
        if (snapshot) { snapshot[c] = nm; snapshot[O + c] = nv; }
        if (c == 0) { obs_rms[2 * O] = tot; if (snapshot) snapshot[2 * O] = tot; }
    }
    if (c == 2 * O && ret_rms) {
        const double bm = l[0][2 * O] / batch_count;
        double bv = l[0][2 * O + 1] / batch_count - bm * bm;
        if (bv < 0.0) bv = 0.0;
        const double mean = ret_rms_in[0], var = ret_rms_in[1], delta = bm - mean, tot = ret_count + batch_count;
        const double nm = mean + delta * batch_count / tot;
        const double nv = (var * ret_count + bv * batch_count + delta * delta * ret_count * batch_count / tot) / tot;
        ret_rms[0] = nm; ret_rms[1] = nv; ret_rms[2] = tot;
        if (snapshot) { snapshot[41] = nm; snapshot[42] = nv; snapshot[43] = tot; }
    }
}

hipError_t launch_moments_merge_tiles(int64_t tiles, const double* partials, double batch_count, const double* obs_rms_in,
                                      const double* ret_rms_in, double* obs_rms, double* ret_rms, double* snapshot,
                                      double* scratch, int obs_words, hipStream_t s) {
    const double* rows = partials;
    int nrows = (int)tiles;
    const int P = 2 * obs_words + 4;
    if (tiles > 1024) {
        const int groups = 256;
        const int64_t per_group = (tiles + groups - 1) / groups;
        hipLaunchKernelGGL(tile_partials_reduce_kernel, dim3(groups), dim3(256), 0, s, partials, tiles, per_group, scratch, P);
        rows = scratch;
        nrows = (int)((tiles + per_group - 1) / per_group);
    }
    hipLaunchKernelGGL(tile_partials_merge_kernel, dim3(1), dim3(1024), 0, s, rows, nrows, batch_count,
                       obs_rms_in ? obs_rms_in : obs_rms, ret_rms_in ? ret_rms_in : ret_rms, obs_rms, ret_rms, snapshot,
                       obs_words);
    return hipGetLastError();
}

// r_norm[t][i] = clip(r[t][i] / sqrt(var_t + eps)) with the return variance as it stood after step t
__global__ __launch_bounds__(256) void reward_finalize_kernel(int64_t T, int64_t n, const float* __restrict__ raw,
                                                              const double* __restrict__ snapshots, float clip, float eps,
                                                              float* __restrict__ out) {
    const int64_t t = blockIdx.y;
    const float inv = (float)(1.0 / sqrt(snapshots[t * kPartialWords + 42] + (double)eps));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[t * n + i] = fminf(fmaxf(raw[t * n + i] * inv, -clip), clip);
}

hipError_t launch_reward_finalize(int64_t T, int64_t n, const float* raw, const double* snapshots, float clip, float eps,
                                  float* out, hipStream_t s) {
    int64_t bx = (n + 256 * 4 - 1) / (256 * 4);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(reward_finalize_kernel, dim3((unsigned)bx, (unsigned)T), dim3(256), 0, s, T, n, raw, snapshots, clip,
                       eps, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- timeout bootstrap without a host round trip
// SB3's on-policy collector adds gamma * V(terminal observation) to the reward of an episode end that is a truncation
// and not a termination (stable_baselines3 OnPolicyAlgorithm.collect_rollouts; the truncation of hover.py:275).  The
// episode ends of a rollout sit in the per-tile done ledger.  ledger_scan walks every ledger slot, counts episodes
// (Monitor statistics), and compacts the truncated-not-terminated entries — terminal observation normalised with the
// statistics of ITS step — into a dense list; the V net then runs over that list with a device-side row count and
// ledger_scatter adds gamma * V.  The list order depends on atomics, the result does not: every entry lands on its own
// (step, drone) cell.
__global__ void zero_words_kernel(uint32_t* __restrict__ a, int64_t na, uint32_t* __restrict__ b, int64_t nb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0u;
        else b[i - na] = 0u;
    }
}

// Start of a rollout in one launch: clear the done ledger's per-tile counts and carry the episode-start flags over
// (dst[i] = src[i], n bytes; a memcpy node in a captured graph costs tens of microseconds of dispatch latency, a kernel node ~1.5).
__global__ void rollout_begin_kernel(uint32_t* __restrict__ counts, int64_t tiles, uint8_t* __restrict__ dst,
                                     const uint8_t* __restrict__ src, int64_t n, int32_t* __restrict__ scan_counters,
                                     double* __restrict__ scan_sums) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = i; k < tiles; k += stride) counts[k] = 0u;
    if (scan_counters && i < 4) scan_counters[i] = 0;    // what this rollout's ledger scan accumulates into
    if (scan_sums && i >= 4 && i < 6) scan_sums[i - 4] = 0.0;
    if (dst) {
        const int64_t words = n / 16;                  // both rows are 16-byte aligned (checked by the caller)
        for (int64_t k = i; k < words; k += stride) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
        for (int64_t k = words * 16 + i; k < n; k += stride) dst[k] = src[k];
    }
}

hipError_t launch_rollout_begin(uint32_t* counts, int64_t tiles, uint8_t* dst, const uint8_t* src, int64_t n, int32_t* scan_counters,
                                double* scan_sums, hipStream_t s) {
    int64_t work = tiles > n / 16 ? tiles : n / 16;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rollout_begin_kernel, dim3((unsigned)blocks), dim3(256), 0, s, counts, tiles, dst, src, n, scan_counters, scan_sums);
    return hipGetLastError();
}

hipError_t launch_zero_words(uint32_t* a, int64_t n, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, n, (uint32_t*)nullptr, (int64_t)0);
    return hipGetLastError();
}

// ---------------------------------------------------------------- the done ledger as ONE dense block (host-mode VecEnv)
// A host that reads the ledger after every step (FpvVecEnv's numpy mode: SB3's infos[i]["terminal_observation"], ["episode"],
// hover.py:102's flags) wants one copy, not a count, a compaction and five gathers.  One workgroup: exclusive prefix of the tiles'
// row counts (chunks of 1 024 tiles, a running base), then every thread copies its tile's rows — a step ends ~N / 31 episodes, a
// couple per tile — as {idx, info, tag, ep_return, obs[O]} to out + 8 + (4 + O) * k, in (tile, row) order, i.e. by drone index for
// a ledger that holds one step.  out[0] = rows appended since the clear (rows that did not fit included), out[1] = rows packed,
// out[2] = row words.  With rms the terminal observation is normalised as normalize_apply_kernel does it (VecNormalize.step_wait
// normalises infos[i]["terminal_observation"] with the statistics as they stand after the step's update).
__global__ __launch_bounds__(1024) void ledger_pack_kernel(const uint32_t* __restrict__ led_count, const int32_t* __restrict__ led_idx,
                                                           const uint32_t* __restrict__ led_info, const uint32_t* __restrict__ led_tag,
                                                           const float* __restrict__ led_ret, const float* __restrict__ led_obs,
                                                           int64_t tiles, int rows, int O, const double* __restrict__ rms, float clip,
                                                           float eps, uint32_t* __restrict__ out, int64_t max_rows) {
    __shared__ uint32_t wsum[16];
    __shared__ float mu[32], inv[32];
    __shared__ uint32_t base_s, total_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { base_s = 0u; total_s = 0u; }
    if (tid < O) {
        mu[tid] = rms ? (float)rms[tid] : 0.0f;
        inv[tid] = rms ? (float)(1.0 / sqrt(rms[O + tid] + (double)eps)) : 1.0f;
    }
    __syncthreads();
    const int W = 4 + O;
    for (int64_t t0 = 0; t0 < tiles; t0 += 1024) {
        const int64_t t = t0 + tid;
        const uint32_t appended = t < tiles ? led_count[t] : 0u;
        const uint32_t c = appended < (uint32_t)rows ? appended : (uint32_t)rows;
        uint32_t incl = c, app = appended;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) app += __shfl_xor(app, d);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = base_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (lane == 0) atomicAdd(&total_s, app);
        const uint32_t first = before + incl - c;
        for (uint32_t r = 0; r < c; ++r) {
            const int64_t k = (int64_t)first + r;
            if (k >= max_rows) break;
            const int64_t slot = t * rows + r;
            uint32_t* row = out + 8 + k * W;
            row[0] = (uint32_t)led_idx[slot];
            row[1] = led_info[slot];
            row[2] = led_tag[slot];
            row[3] = __float_as_uint(led_ret[slot]);
            for (int j = 0; j < O; ++j) {
                const float x = led_obs[slot * O + j];
                row[4 + j] = __float_as_uint(rms ? fminf(fmaxf((x - mu[j]) * inv[j], -clip), clip) : x);
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t s = 0u;
            for (int w = 0; w < 16; ++w) s += wsum[w];
            base_s += s;
        }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = total_s;
        out[1] = (uint32_t)((int64_t)base_s < max_rows ? (int64_t)base_s : max_rows);
        out[2] = (uint32_t)W;
        out[3] = base_s;
        out[4] = out[5] = out[6] = out[7] = 0u;
    }
}

hipError_t launch_ledger_pack(const uint32_t* led_count, const int32_t* led_idx, const uint32_t* led_info, const uint32_t* led_tag,
                              const float* led_ret, const float* led_obs, int64_t tiles, int rows, int obs_words, const double* rms,
                              float clip, float eps, uint32_t* out, int64_t max_rows, hipStream_t s) {
    hipLaunchKernelGGL(ledger_pack_kernel, dim3(1), dim3(1024), 0, s, led_count, led_idx, led_info, led_tag, led_ret, led_obs, tiles,
                       rows, obs_words, rms, clip, eps, out, max_rows);
    return hipGetLastError();
}

struct LedgerScanArgs {
    const uint32_t* led_count;
    const int32_t* led_idx;
    const uint32_t* led_info;
    const uint32_t* led_tag;
    const float* led_ret;
    const float* led_obs;
    const uint64_t* tick;     // the handle's tile-0 agent tick AFTER the rollout
    const double* snapshots;  // [T][stride]: {mean[O], var[O], ...} after each step, or null (no normalisation)
    int64_t tiles, stride, cap;
    int32_t rows, T, obs_words, table;   // table: the per-step normaliser is tabulated in dynamic LDS (T * obs_words * 16 bytes)
    float clip_obs, eps;
    float* tobs;              // [cap][O]
    int32_t* t_idx;           // [cap]
    int32_t* d_idx;           // [cap]
    int32_t* counters;        // {K, valid rows, episodes (incl. overflowed), flags: 1 ledger overflow, 2 tag outside rollout, 4 list overflow}
    double* sums;             // {sum ep_return, sum ep_len}
    uint32_t* ticket;         // per-handle device word, zero between launches (used when status_out is given)
    void* status_out;         // optional, 32 bytes {counters[4], sums[2]} as they stand when every workgroup has added its share;
                              // may be host memory the device can write (pinned): the host then needs no copy behind the launch
    // fpv_ledger_scan_finalize: the rollout's deferred reward scaling (reward_finalize_kernel's arithmetic) rides in this launch —
    // the workgroups behind the first scan_blocks take it — instead of being a launch of its own in front of it
    const float* fin_raw;     // [T][fin_n] or null
    float* fin_out;
    const double* fin_snap;   // [T][fin_stride]: word 42 = the return variance after step t
    int64_t fin_n, fin_stride;
    float fin_clip, fin_eps;
    int32_t scan_blocks;      // workgroups 0 .. scan_blocks - 1 walk the ledger
};

// Each workgroup walks kScanPerThread x 256 consecutive slots and adds its statistics with ONE set of atomics (an
// untrained policy ends ~N episodes per rollout: one atomic set per wave cost 70 us at 65 536 drones).
constexpr int kScanPerThread = 16;

__global__ __launch_bounds__(256) void ledger_scan_kernel(LedgerScanArgs a) {
    __shared__ double red_d[4][2];
    __shared__ int red_i[4][3];
    extern __shared__ double nrm_lds[];                          // [T][obs_words]{mean, 1 / sqrt(var + eps)} when a.table != 0
    if ((int)blockIdx.x >= a.scan_blocks) {                      // (uniform over the workgroup)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        // r_norm[t][i] = clip(r[t][i] / sqrt(var_t + eps)), var_t = snapshots[t][42]: reward_finalize_kernel on this launch's spare
        // workgroups; 16 bytes per lane and trip
        // ([T][n] is one flat array: the workgroups stride over ALL of it, the step of an element is its index / n)
        const int64_t total = (int64_t)a.T * a.fin_n, fb = (int64_t)blockIdx.x - a.scan_blocks, nfb = (int64_t)gridDim.x - a.scan_blocks;
        const bool vec = (a.fin_n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.fin_raw) | reinterpret_cast<uintptr_t>(a.fin_out)) & 15) == 0;
        const int64_t quads = vec ? total / 4 : 0;
        const f32x4* src = reinterpret_cast<const f32x4*>(a.fin_raw);
        f32x4* dst = reinterpret_cast<f32x4*>(a.fin_out);
        int64_t t_have = -1;
        float inv = 0.f;
        for (int64_t q = fb * 256 + threadIdx.x; q < quads; q += nfb * 256) {
            const int64_t t = (4 * q) / a.fin_n;                 // (n % 4 == 0: a quad never straddles two steps)
            if (t != t_have) { inv = (float)(1.0 / sqrt(a.fin_snap[t * a.fin_stride + 42] + (double)a.fin_eps)); t_have = t; }
            const f32x4 v = src[q];
            dst[q] = f32x4{fminf(fmaxf(v.x * inv, -a.fin_clip), a.fin_clip), fminf(fmaxf(v.y * inv, -a.fin_clip), a.fin_clip),
                           fminf(fmaxf(v.z * inv, -a.fin_clip), a.fin_clip), fminf(fmaxf(v.w * inv, -a.fin_clip), a.fin_clip)};
        }
        for (int64_t i = quads * 4 + fb * 256 + threadIdx.x; i < total; i += nfb * 256) {
            const int64_t t = i / a.fin_n;
            if (t != t_have) { inv = (float)(1.0 / sqrt(a.fin_snap[t * a.fin_stride + 42] + (double)a.fin_eps)); t_have = t; }
            a.fin_out[i] = fminf(fmaxf(a.fin_raw[i] * inv, -a.fin_clip), a.fin_clip);
        }
        return;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t slots = a.tiles * a.rows;
    const uint32_t tick0 = (uint32_t)(a.tick[0] - (uint64_t)a.T);
    double sr = 0.0, sl = 0.0;
    int nv = 0, episodes = 0, flags = 0;
    // Two rounds of independent loads instead of kScanPerThread dependent ones: every slot's tile count first, then — with the
    // index clamped to slot 0 where the slot is empty, so that no load sits under a run-time test — info / return / tag.
    uint32_t cnt_[kScanPerThread], info_[kScanPerThread], tag_[kScanPerThread];
    float ret_[kScanPerThread];
    __shared__ uint32_t wave_list[4][kScanPerThread * 64];   // per wave: its truncated-not-terminated entries (16 KB)
    int wave_total = 0;              // ... and how many there are (uniform)
#pragma unroll
    for (int it = 0; it < kScanPerThread; ++it) {
        const int64_t s = ((int64_t)blockIdx.x * kScanPerThread + it) * 256 + threadIdx.x;
        const int tile = s < slots ? (int)((uint32_t)s / (uint32_t)a.rows) : 0;     // slots < 2^31 (checked by the launcher)
        cnt_[it] = a.led_count[tile];
    }
#pragma unroll
    for (int it = 0; it < kScanPerThread; ++it) {
        const int64_t s = ((int64_t)blockIdx.x * kScanPerThread + it) * 256 + threadIdx.x;
        const int tile = s < slots ? (int)((uint32_t)s / (uint32_t)a.rows) : 0;
        const int row = (int)s - tile * a.rows;
        const uint32_t lim = cnt_[it] < (uint32_t)a.rows ? cnt_[it] : (uint32_t)a.rows;
        const int64_t q = (s < slots && (uint32_t)row < lim) ? s : 0;
        info_[it] = a.led_info[q]; ret_[it] = a.led_ret[q]; tag_[it] = a.led_tag[q];
    }
#pragma unroll
    for (int it = 0; it < kScanPerThread; ++it) {
        const int64_t s = ((int64_t)blockIdx.x * kScanPerThread + it) * 256 + threadIdx.x;
        bool trunc = false;
        int32_t t = 0;
        if (s < slots) {
            const int tile = (int)((uint32_t)s / (uint32_t)a.rows);
            const int row = (int)s - tile * a.rows;
            const uint32_t cnt = cnt_[it];
            if (row == 0) {
                episodes += (int)cnt;
                if (cnt > (uint32_t)a.rows) flags |= 1;
            }
            if ((uint32_t)row < (cnt < (uint32_t)a.rows ? cnt : (uint32_t)a.rows)) {
                const uint32_t info = info_[it];
                ++nv;
                sr += (double)ret_[it];
                sl += (double)(info >> 8);
                t = (int32_t)(tag_[it] - tick0);
                trunc = (info & kInfoTrunc) && !(info & kInfoTerm);
                if (trunc && (t < 0 || t >= a.T)) { flags |= 2; trunc = false; }
            }
        }
        // this wave's truncated-not-terminated entries, in (round, lane) order: (slot within the workgroup's range) << 16 | step
        const unsigned long long m = __ballot(trunc);
        if (trunc) wave_list[wv][wave_total + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)(it * 256 + (int)threadIdx.x) << 16) | (uint32_t)t;
        wave_total += __popcll(m);
    }
    // truncated-not-terminated entries -> dense list.  ONE reservation per wave (an atomic with a return per round of the loop
    // above cost a cross-XCD round trip each), and the wave's entries are then worked through from its LDS list, every lane busy:
    // first {t_idx, d_idx} of 64 entries at a time, then the terminal observations — normalised and copied one lane per word,
    // E = 64 / obs_words entries side by side per pass, two passes per memory round trip, every load requested from an address
    // that is always valid before the first is used.  (Entry after entry, round after round, this was 35 us at 65 536 yaw drones,
    // where ~3 300 episodes per 20-step rollout end by truncation.)  mean and 1 / sqrt(var + eps) of every step come from the
    // table in LDS (above) when it fits.
    // The per-step normaliser table is built only by a workgroup that has such entries (a 65 536-drone hover rollout of an
    // untrained policy has next to none: episodes end by termination), and behind the statistics pass, whose loads it would
    // otherwise stand in front of.
    const double* nrm = nullptr;
    if (a.snapshots && a.table && __syncthreads_or(wave_total)) {
        for (int e = threadIdx.x; e < a.T * a.obs_words; e += 256) {
            const int tt = e / a.obs_words, w = e - tt * a.obs_words;
            nrm_lds[2 * e] = a.snapshots[(int64_t)tt * a.stride + w];
            nrm_lds[2 * e + 1] = 1.0 / sqrt(a.snapshots[(int64_t)tt * a.stride + a.obs_words + w] + (double)a.eps);
        }
        nrm = nrm_lds;
        __syncthreads();
    }
    if (wave_total) {                                                  // uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int base = 0;
        if (lane == 0) base = atomicAdd(&a.counters[0], wave_total);
        base = __shfl(base, 0);
        const int64_t slot0 = (int64_t)blockIdx.x * kScanPerThread * 256;
        const int fit = (int64_t)base >= a.cap ? 0 : (int)(((int64_t)base + wave_total <= a.cap) ? wave_total : a.cap - base);
        if (fit < wave_total) flags |= 4;
        for (int k = lane; k < fit; k += 64) {
            const uint32_t ent = wave_list[wv][k];
            a.t_idx[base + k] = (int32_t)(ent & 0xffffu);
            a.d_idx[base + k] = a.led_idx[slot0 + (ent >> 16)];
        }
        const int O = a.obs_words, E = 64 / O;
        const int g = lane / O, w = lane - g * O;
        for (int k0 = 0; k0 < fit; k0 += 2 * E) {
            int kk[2], tb[2];
            float x[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                kk[u] = k0 + u * E + g;
                if (g >= E || kk[u] >= fit) kk[u] = -1;
                const uint32_t ent = wave_list[wv][kk[u] < 0 ? 0 : kk[u]];
                tb[u] = (int)(ent & 0xffffu);
                x[u] = a.led_obs[(slot0 + (ent >> 16)) * O + w];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (kk[u] < 0) continue;
                float y = x[u];
                if (a.snapshots) {
                    double mean, inv;
                    if (nrm) {
                        mean = nrm[(tb[u] * O + w) * 2];
                        inv = nrm[(tb[u] * O + w) * 2 + 1];
                    } else {
                        mean = a.snapshots[(int64_t)tb[u] * a.stride + w];
                        inv = 1.0 / sqrt(a.snapshots[(int64_t)tb[u] * a.stride + O + w] + (double)a.eps);
                    }
                    const double z = ((double)y - mean) * inv;
                    y = (float)fmin(fmax(z, -(double)a.clip_obs), (double)a.clip_obs);
                }
                a.tobs[(int64_t)(base + kk[u]) * O + w] = y;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sr += __shfl_xor(sr, off); sl += __shfl_xor(sl, off);
        nv += __shfl_xor(nv, off); episodes += __shfl_xor(episodes, off); flags |= __shfl_xor(flags, off);
    }
    if (lane == 0) { red_d[wv][0] = sr; red_d[wv][1] = sl; red_i[wv][0] = nv; red_i[wv][1] = episodes; red_i[wv][2] = flags; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            sr += red_d[w][0]; sl += red_d[w][1]; nv += red_i[w][0]; episodes += red_i[w][1]; flags |= red_i[w][2];
        }
        if (nv) { atomicAdd(&a.counters[1], nv); atomicAdd(&a.sums[0], sr); atomicAdd(&a.sums[1], sl); }
        if (episodes) atomicAdd(&a.counters[2], episodes);
        if (flags) atomicOr(&a.counters[3], flags);
        if (a.status_out) {
            // The workgroup that draws the last ticket sees every other workgroup's additions (each fenced its own before taking
            // a ticket) and publishes the totals with system-scope stores: a D2H copy node behind the rollout's graph stood
            // ~20 us between one rollout's last kernel and the next one's first (profiles/r03_rollout_boundary.md).
            __threadfence();
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (tk == (unsigned)a.scan_blocks - 1u) {
                int32_t* oi = reinterpret_cast<int32_t*>(a.status_out);
                double* od = reinterpret_cast<double*>(a.status_out) + 2;
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(oi + k, __hip_atomic_load(a.counters + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int k = 0; k < 2; ++k)
                    __hip_atomic_store(od + k, __hip_atomic_load(a.sums + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(256) void ledger_scatter_kernel(const int32_t* __restrict__ counters, const int32_t* __restrict__ t_idx,
                                                             const int32_t* __restrict__ d_idx, const float* __restrict__ v,
                                                             float gamma, float* __restrict__ rewards, int64_t n, int64_t cap) {
    int64_t k_end = counters[0];
    if (k_end > cap) k_end = cap;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < k_end; k += (int64_t)gridDim.x * 256)
        rewards[(int64_t)t_idx[k] * n + d_idx[k]] += gamma * v[k];
}

hipError_t launch_ledger_scan(const uint32_t* led_count, const int32_t* led_idx, const uint32_t* led_info, const uint32_t* led_tag,
                              const float* led_ret, const float* led_obs, const uint64_t* tick, int64_t tiles, int rows,
                              int obs_words, const double* snapshots, int64_t stride, int T, float clip_obs, float eps,
                              float* tobs, int32_t* t_idx, int32_t* d_idx, int64_t cap, int32_t* counters, double* sums,
                              bool counters_are_zero, uint32_t* ticket, void* status_out, const float* fin_raw, float* fin_out,
                              const double* fin_snap, int64_t fin_stride, int64_t fin_n, float fin_clip, float fin_eps, hipStream_t s) {
    // zeroed by a kernel — this one, unless the rollout's fpv_rollout_begin has done it — never by hipMemsetAsync: a memset node
    // in a captured graph may write garbage from the graph's second replay on (profiles/r03_graph_memset_probe.json)
    if (!counters_are_zero)
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<uint32_t*>(counters), (int64_t)4,
                           reinterpret_cast<uint32_t*>(sums), (int64_t)4);
    LedgerScanArgs a{led_count, led_idx, led_info, led_tag, led_ret, led_obs, tick, snapshots, tiles, stride, cap,
                     rows, T, obs_words, 0, clip_obs, eps, tobs, t_idx, d_idx, counters, sums, ticket, status_out,
                     fin_raw, fin_out, fin_snap, fin_n, fin_stride, fin_clip, fin_eps, 0};
    const int64_t slots = tiles * rows, per_block = 256 * kScanPerThread;
    if (slots >= (1ll << 31) || cap >= (1ll << 31) || T > 65535) return hipErrorInvalidValue;   // (the step rides in 16 bits of a list entry)
    const size_t table_bytes = (size_t)T * (size_t)obs_words * 16;
    a.table = snapshots && table_bytes <= 40 * 1024 ? 1 : 0;        // (longer rollouts: the normaliser is computed per entry; 16 KB of
                                                                    // the 64 KB a workgroup gets without asking are the waves' lists)
    a.scan_blocks = (int32_t)((slots + per_block - 1) / per_block);
    int64_t fin_blocks = 0;
    if (fin_raw) {                                   // 16 KB of rewards per workgroup and trip, at most 512 workgroups
        fin_blocks = ((int64_t)T * fin_n + 4095) / 4096;
        if (fin_blocks > 512) fin_blocks = 512;
        if (fin_blocks < 1) fin_blocks = 1;
    }
    hipLaunchKernelGGL(ledger_scan_kernel, dim3((unsigned)(a.scan_blocks + fin_blocks)), dim3(256),
                       a.table ? table_bytes : 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ledger_scatter(const int32_t* counters, const int32_t* t_idx, const int32_t* d_idx, const float* v, float gamma,
                                 float* rewards, int64_t n, int64_t cap, hipStream_t s) {
    int64_t blocks = (cap + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(ledger_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, s, counters, t_idx, d_idx, v, gamma, rewards, n, cap);
    return hipGetLastError();
}

hipError_t launch_normalize_reward(int64_t n, const float* r, const uint8_t* done, float* ret, float* r_out,
                                   double* rms, int32_t update, float gamma, float clip, float eps, double* scratch,
                                   hipStream_t s) {
    int64_t blocks = (n + kRedBlock - 1) / kRedBlock;
    if (blocks > kMaxRedBlocks) blocks = kMaxRedBlocks;
    if (update) {
        hipLaunchKernelGGL(return_update_kernel, dim3((unsigned)blocks), dim3(kRedBlock), 0, s, n, r, ret, gamma,
                           scratch);
        // a [blocks][2] partial table is the d = 1 case of the obs merge
        hipLaunchKernelGGL(moments_merge_kernel, dim3(1), dim3(32 * kMergeWays), 0, s, scratch, (int)blocks, 1, (double)n,
                           rms);
    }
    hipLaunchKernelGGL(reward_apply_kernel, dim3((unsigned)blocks), dim3(kRedBlock), 0, s, n, r, done, ret, r_out, rms,
                       clip, eps);
    return hipGetLastError();
}

}  // namespace fpv
