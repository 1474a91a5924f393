// fpv_policy.hip — fused actor-critic forward of the PPO rollout (gfx950, exact-fp32 MFMA).
//
// One kernel does what SB3's MlpPolicy.forward + VecNormalize.normalize_obs + Box clip do per
// rollout step (simulation/train_hover.py:43-58, net_arch=[128,128]):
//   raw obs [N][20] -> (x - mean) / sqrt(var + eps), clip        (VecNormalize statistics, fp64 in)
//   pi: 20 -> 128 -> 128 -> 4 (tanh), V: 20 -> 128 -> 128 -> 1 (tanh)
//   a = mu + exp(log_std) * eps,  log pi(a),  clip(a, -1, 1),  V
// Layout: everything is computed TRANSPOSED, H^T = W . X^T, on v_mfma_f32_32x32x2_f32 (bit-exact
// fp32 fma chains, MI355X_MICROARCH.md): drones sit on the MFMA column (= lane & 31), the 32x32
// accumulator tile of one layer has its hidden index in the 16 registers, so it is the next layer's
// B operand as it stands — no LDS round trip, no lane movement between layers.  The K order that
// this induces is baked into the weight image by pack_kernel.  A 512-thread workgroup keeps the
// whole 160 KB weight image in LDS and walks over 256-drone chunks (32 drones per wave).
#include "fpv_mlp.h"
#include "fpv_policy_image.h"

namespace fpv {

// ---------------------------------------------------------------- weight image
// nn.Linear layouts: w1 [128][20], w2 [128][128], wa [4][128], wv [1][128] (row = output).
struct PolicyParams {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
    int in_dim, act_dim;   // 20 / 4 hover (hover.py:59-70), 12 / 1 yaw (yaw.py:37-45); in_dim % 4 == 0, <= 24; act_dim <= 4
                           // layer 1 takes in_dim / 2 products per tile: lane half h supplies k = s + (in_dim / 2) h, no padding
};

__global__ void policy_pack_kernel(PolicyParams p, float* __restrict__ img) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kImageFloats) return;
    float v = 0.0f;
    if (e < kHeadBase) {
        const int net = e / kNetFloats, o = e % kNetFloats;
        const float* w1 = net ? p.vf_w1 : p.pi_w1;
        const float* b1 = net ? p.vf_b1 : p.pi_b1;
        const float* w2 = net ? p.vf_w2 : p.pi_w2;
        const float* b2 = net ? p.vf_b2 : p.pi_b2;
        if (o < kOffB1) {                       // W1p[m][s][lane] = W1[m*32 + i][s + (in_dim / 2) h], s < in_dim / 2
            const int lane = o % 64, s = (o / 64) % 12, m = o / (64 * 12);
            const int half = p.in_dim / 2, k = s + half * (lane >> 5);
            v = s < half ? w1[(m * 32 + (lane & 31)) * p.in_dim + k] : 0.0f;
        } else if (o < kOffW2) {                // b1p[m][h][r]
            const int q = o - kOffB1, r = q % 16, h = (q / 16) % 2, m = q / 32;
            v = b1[m * 32 + tile_row(r, h)];
        } else if (o < kOffB2) {                // W2p[m][kt][r4][lane][rr] = W2[m*32 + i][kt*32 + row(4 r4 + rr, h)]
            const int q = o - kOffW2, rr = q % 4, lane = (q / 4) % 64, r4 = (q / 256) % 4, kt = (q / 1024) % 4, m = q / 4096;
            v = w2[(m * 32 + (lane & 31)) * kHid + kt * 32 + tile_row(4 * r4 + rr, lane >> 5)];
            if (FPV_TANH_FOLD) v *= -2.0f;      // layer 2 consumes r1 = (1 - tanh) / 2 (fpv_mlp.h)
        } else {
            const int q = o - kOffB2, r = q % 16, h = (q / 16) % 2, m = q / 32;
            const int j = m * 32 + tile_row(r, h);
            double b = (double)b2[j];
            if (FPV_TANH_FOLD)
                for (int k = 0; k < kHid; ++k) b += (double)w2[j * kHid + k];
            v = (float)b;
        }
        v *= kTanhPrescale;                     // both hidden layers end in tanh: tanh_carrier_prescaled (fpv_mlp.h)
    } else {
        const int q = e - kHeadBase;
        if (q < 512) {                          // Wa[h][kt][r][o]
            const int o = q % 4, r = (q / 4) % 16, kt = (q / 64) % 4, h = q / 256;
            v = o < p.act_dim ? p.wa[o * kHid + kt * 32 + tile_row(r, h)] : 0.0f;
            if (FPV_TANH_FOLD) v *= -2.0f;      // the heads consume r2
        } else if (q < 640) {                   // Wv[h][kt][r]
            const int z = q - 512, r = z % 16, kt = (z / 16) % 4, h = z / 64;
            v = p.wv[kt * 32 + tile_row(r, h)];
            if (FPV_TANH_FOLD) v *= -2.0f;
        } else if (q < 644) {
            const int o = q - 640;
            double b = o < p.act_dim ? (double)p.ba[o] : 0.0;
            if (FPV_TANH_FOLD && o < p.act_dim)
                for (int k = 0; k < kHid; ++k) b += (double)p.wa[o * kHid + k];
            v = (float)b;
        } else if (q == 644) {
            double b = (double)p.bv[0];
            if (FPV_TANH_FOLD)
                for (int k = 0; k < kHid; ++k) b += (double)p.wv[k];
            v = (float)b;
        }
        else if (q >= 648 && q < 652) v = q - 648 < p.act_dim ? p.log_std[q - 648] : 0.0f;
        else if (q >= 652 && q < 656) v = q - 652 < p.act_dim ? expf(p.log_std[q - 652]) : 0.0f;
        else if (q == 656) {                    // -sum log_std - act_dim * 0.5 ln 2 pi
            for (int j = 0; j < p.act_dim; ++j) v -= p.log_std[j] + 0.5f * 1.8378770664093453f;
        }
    }
    img[e] = v;
}

struct PolicyArgs {
    const float* image;      // packed weights (kImageFloats)
    const float* obs;        // [n][20] raw observations
    const double* rms;       // mean[20], var[20], count — or null: observations are used as they are
    const float* eps_in;     // [n][4] standard normal draws, or null: Philox in-kernel
    const uint64_t* tick;    // device counter keying the in-kernel RNG (the env's per-tile tick), or null
    float* obs_n;            // [n][20] normalised observation (rollout buffer), may be null
    float* action;           // [n][4] sampled action (unclipped), may be null
    float* clipped;          // [n][4] clip(action, -1, 1) — what the env receives, may be null
    float* value;            // [n]
    float* logp;             // [n], may be null
    float* eps_out;          // [n][4], may be null
    int64_t n;
    uint64_t seed, id_offset;
    float clip_obs, eps_norm;
    int32_t deterministic;   // 1: action = mean (SB3 predict(deterministic=True), test_hover.py:17)
    int32_t in_dim, act_dim;
    MergeArgs mg;            // mg.rows != null: fold the previous env step's tile sums into rms first (512 threads)
    const int32_t* n_dev;    // optional device-side row count: rows = min(n, *n_dev)
    const int32_t* sc_t;     // optional: sc_out[sc_t[d] * sc_stride + sc_d[d]] += sc_scale * value[d] (fpv_policy_io.scatter_*)
    const int32_t* sc_d;
    float* sc_out;
    int64_t sc_stride;
    float sc_scale;
};

// THREADS = 512: two waves per SIMD (the stand-alone shape).  THREADS = 256: one wave per SIMD and <= 256 VGPRs of
// the 512-entry file, which leaves room for the VALU-bound env-step waves to co-reside: the critic pass is run that
// way on a second stream beside fpv_step (matrix pipe and vector pipe are separate on CDNA4).
// Nets [net_first, net_first + net_count) of {0: pi, 1: V} are evaluated; only their images are staged into LDS.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void policy_forward_kernel(PolicyArgs p, int net_first, int net_count) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* hb = lds + net_count * kNetFloats;        // heads [660], then normaliser mean[20], inv[20]
    // ---- stage the weight image (L2-resident after the first workgroups) and the normaliser: every granule of a
    // thread is requested before the first one is written to LDS, so the block pays one L2 round trip
    FPV_STAMP(0);
    // the first chunk's observation rows are requested ahead of the weight image: their latency (the rows were just written
    // by the env step, on other XCDs) then hides under the 160 KB of staging instead of following it (4.9 k cycles of the
    // launch in profiles/r02_policy_phase_stamps.md)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int half = p.in_dim / 2;                   // words of a row held by one lane half: k = half * h + j
    f32x2 xpre[6];
    {
        const int64_t d0 = (int64_t)blockIdx.x * (THREADS / 2) + (threadIdx.x >> 6) * 32 + (threadIdx.x & 31);
        const int hh = (threadIdx.x >> 5) & 1;
        const f32x2* row = reinterpret_cast<const f32x2*>(p.obs + (d0 < p.n ? d0 : 0) * p.in_dim + half * hh);
#pragma unroll
        for (int q = 0; q < 6; ++q) xpre[q] = row[2 * q < half ? q : 0];
    }
    MergeRegs mr;
    if (p.mg.rows) merge_request(p.mg, mr);     // the tile sums travel together with the weight image
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.image + (size_t)net_first * kNetFloats);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        constexpr int kNetGran = kNetFloats / 4, kPer = (kNetGran + THREADS - 1) / THREADS;
        constexpr int kHeadGran = (kImageFloats - kHeadBase) / 4;   // 165 granules, requested together with the first net
        const f32x4 hv = reinterpret_cast<const f32x4*>(p.image + kHeadBase)[threadIdx.x < kHeadGran ? threadIdx.x : 0];
        for (int sl = 0; sl < net_count; ++sl) {
            f32x4 stage[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * THREADS + threadIdx.x;
                stage[q] = src[sl * kNetGran + (g < kNetGran ? g : 0)];
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * THREADS + threadIdx.x;
                if (g < kNetGran) dst[sl * kNetGran + g] = stage[q];
            }
        }
        if (threadIdx.x < kHeadGran) reinterpret_cast<f32x4*>(hb)[threadIdx.x] = hv;
        if (!p.mg.rows && (int)threadIdx.x < p.in_dim) {
            hb[660 + threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
            hb[680 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[p.in_dim + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
        }
    }
    FPV_STAMP(1);
    if (p.mg.rows)                              // uniform: the normaliser comes out of the merged statistics
        merge_prologue(p.mg, mr, p.rms, reinterpret_cast<double*>(hb + 700), p.eps_norm, hb + 660, hb + 680);
    __syncthreads();
    FPV_STAMP(2);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const float clip = p.rms ? p.clip_obs : 3.0e38f;
    const bool do_pi = net_first == 0, do_v = net_first + net_count == 2;
    constexpr int kChunk = THREADS / 2;              // 32 drones per wave

    int64_t rows = p.n;
    int chunk = kChunk;                              // rows per workgroup and pass: 32 per wave
    if (p.n_dev) {
        // A list whose length only the device knows (the truncated episodes of a rollout) is usually short.  A workgroup takes as
        // long for 256 rows as the whole grid for 256 each, so the rows are spread: 32 per wave, and only as many waves per
        // workgroup as it takes to cover the list with the whole grid (3 300 rows at 65 536 yaw drones: 24 -> 9 us).
        const int64_t nd = *p.n_dev;
        rows = nd < rows ? (nd > 0 ? nd : 0) : rows;
        const int64_t waves = (rows + 32 * (int64_t)gridDim.x - 1) / (32 * (int64_t)gridDim.x);
        const int wu = waves < 1 ? 1 : (waves > THREADS / 64 ? THREADS / 64 : (int)waves);
        if (wave >= wu) return;                      // (no workgroup barrier from here on)
        chunk = 32 * wu;
    }
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < rows; base += (int64_t)gridDim.x * chunk) {
        const int64_t d = base + wave * 32 + col;   // this lane's drone (both halves of the wave share it)
        const bool live = d < rows;
        // ---- observation: half h of the wave holds words k = half * h + j, j < half (10 + 10 for the hover row)
        float x[12];
        {
            const f32x2* row = reinterpret_cast<const f32x2*>(p.obs + (live ? d : 0) * p.in_dim + half * h);
            const bool first_chunk = !p.n_dev && base == (int64_t)blockIdx.x * kChunk;   // its rows were requested before the staging
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const bool have = live && 2 * q < half;
                f32x2 v = xpre[q];
                if (!first_chunk) v = row[have ? q : 0];
                if (!have) v = f32x2{0.f, 0.f};
                x[2 * q] = v.x; x[2 * q + 1] = v.y;
                xpre[q] = f32x2{0.f, 0.f};      // dead from here on
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int k = half * h + j;
                if (j < half) x[j] = fminf(fmaxf((x[j] - hb[660 + k]) * hb[680 + k], -clip), clip);
            }
            if (p.obs_n && live) {
                f32x2* out = reinterpret_cast<f32x2*>(p.obs_n + d * p.in_dim + half * h);
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    if (2 * q < half) out[q] = f32x2{x[2 * q], x[2 * q + 1]};
            }
        }

        FPV_STAMP(3);
        float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // 4 action means + value (partial sums of this lane half)
#pragma unroll 1
        for (int sl = 0; sl < net_count; ++sl) {
            const int net = net_first + sl;
            const float* nb = lds + sl * kNetFloats;
            // ---- layer 1: in_dim -> 128, in_dim / 2 products per tile (10 for the hover row: no padded k)
            Hidden a;
            bias_init(a, nb + kOffB1, h);
#pragma unroll
            for (int s = 0; s < 12; ++s)
                if (s < half) {                   // uniform
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[kOffW1 + (m * 12 + s) * 64 + lane], x[s], a.t[m], 0, 0, 0);
                }
            FPV_STAMP(4 + 5 * sl);
            carrier_all_prescaled(a);
            FPV_STAMP(5 + 5 * sl);
            // ---- layer 2: 128 -> 128; the layer-1 tiles are the B operands as they stand
            Hidden c;
            bias_init(c, nb + kOffB2, h);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 w[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        w[m] = *reinterpret_cast<const f32x4*>(nb + kOffW2 + ((((m * 4 + kt) * 4 + r4) * 64 + lane) * 4));
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            c.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[m][rr], a.t[kt][4 * r4 + rr], c.t[m], 0, 0, 0);
                }
            FPV_STAMP(6 + 5 * sl);
            carrier_all_prescaled(c);
            FPV_STAMP(7 + 5 * sl);
            // ---- heads on the VALU: this lane holds 64 of its drone's 128 hidden values
            if (net == 0) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(hb + (((h * 4 + kt) * 16 + r) * 4));
                        const float v = c.t[kt][r];
                        head[0] += w.x * v; head[1] += w.y * v; head[2] += w.z * v; head[3] += w.w * v;
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) head[4] += hb[512 + (h * 4 + kt) * 16 + r] * c.t[kt][r];
            }
            FPV_STAMP(8 + 5 * sl);
        }
        // ---- combine the two lane halves, add biases
#pragma unroll
        for (int j = 0; j < 5; ++j) head[j] = half_sum(head[j]);
        if (do_v && live && h == 1) {
            const float v = head[4] + hb[644];
            p.value[d] = v;
            if (p.sc_out) p.sc_out[(int64_t)p.sc_t[d] * p.sc_stride + p.sc_d[d]] += p.sc_scale * v;
        }
        if (!do_pi) continue;
        const float mu[4] = {head[0] + hb[640], head[1] + hb[641], head[2] + hb[642], head[3] + hb[643]};

        // ---- sample: eps from memory, or Philox4x32-10 keyed like the env noise (stream tag 'POL1')
        float eps[4] = {0.f, 0.f, 0.f, 0.f};
        if (!p.deterministic) {
            if (p.eps_in) {
                if (p.act_dim == 4) {
                    const f32x4 e = live ? reinterpret_cast<const f32x4*>(p.eps_in)[d] : f32x4{0.f, 0.f, 0.f, 0.f};
                    eps[0] = e.x; eps[1] = e.y; eps[2] = e.z; eps[3] = e.w;
                } else {
                    for (int j = 0; j < p.act_dim; ++j) eps[j] = live ? p.eps_in[d * p.act_dim + j] : 0.f;
                }
            } else {
                const uint64_t t = p.tick ? p.tick[0] : 0ull;
                uint32_t xr[4];
                philox4x32_10((uint32_t)(p.id_offset + (uint64_t)d), (uint32_t)t, (uint32_t)(t >> 32), 0x504F4C31u,
                           (uint32_t)p.seed, (uint32_t)(p.seed >> 32), xr);
                constexpr float k24 = 5.9604644775390625e-8f, m2ln2 = -1.3862943611198906f;
                const float u0 = ((float)(xr[0] >> 8) + 0.5f) * k24, u1 = ((float)(xr[1] >> 8) + 0.5f) * k24;
                const float u2 = ((float)(xr[2] >> 8) + 0.5f) * k24, u3 = ((float)(xr[3] >> 8) + 0.5f) * k24;
                const float ra = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u0)), rb = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u2));
                eps[0] = ra * __builtin_amdgcn_cosf(u1); eps[1] = ra * __builtin_amdgcn_sinf(u1);
                eps[2] = rb * __builtin_amdgcn_cosf(u3); eps[3] = rb * __builtin_amdgcn_sinf(u3);
            }
        }
        float act[4], lp = hb[656];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            act[j] = mu[j] + hb[652 + j] * eps[j];
            if (j < p.act_dim) lp -= 0.5f * eps[j] * eps[j];
        }
        if (live) {
            if (h == 0) {
                if (p.act_dim == 4) {
                    if (p.action) reinterpret_cast<f32x4*>(p.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
                    if (p.clipped)
                        reinterpret_cast<f32x4*>(p.clipped)[d] =
                            f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f),
                                  fminf(fmaxf(act[2], -1.f), 1.f), fminf(fmaxf(act[3], -1.f), 1.f)};
                } else {
                    for (int j = 0; j < p.act_dim; ++j) {
                        if (p.action) p.action[d * p.act_dim + j] = act[j];
                        if (p.clipped) p.clipped[d * p.act_dim + j] = fminf(fmaxf(act[j], -1.f), 1.f);
                    }
                }
            } else {
                if (p.logp) p.logp[d] = lp;
                if (p.eps_out) {
                    if (p.act_dim == 4) reinterpret_cast<f32x4*>(p.eps_out)[d] = f32x4{eps[0], eps[1], eps[2], eps[3]};
                    else for (int j = 0; j < p.act_dim; ++j) p.eps_out[d * p.act_dim + j] = eps[j];
                }
            }
        }
        FPV_STAMP(14);
    }
}

// ---------------------------------------------------------------- launchers
hipError_t launch_policy_pack(const float* const* w /* 13 device pointers in PolicyParams order */, float* image,
                              int in_dim, int act_dim, hipStream_t s) {
    PolicyParams p{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], in_dim, act_dim};
    hipLaunchKernelGGL(policy_pack_kernel, dim3((kImageFloats + 255) / 256), dim3(256), 0, s, p, image);
    return hipGetLastError();
}

hipError_t launch_policy_forward(const float* image, const float* obs, const double* rms, const float* eps_in,
                                 const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                 float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                 float clip_obs, float eps_norm, int deterministic, int max_blocks, int nets, int narrow,
                                 int in_dim, int act_dim, const MergeArgs& merge, const int32_t* n_dev, const ScatterArgs& sc,
                                 hipStream_t s) {
    // nets: 3 = pi + V (one launch), 1 = pi only (with the sampling epilogue), 2 = V only.
    // narrow: 256-thread workgroups (one wave per SIMD) so that other kernels can co-reside.
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    if (!attr_set) {
        const int full = (int)((2 * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float)) + kMergeLdsBytes;
        hipError_t e = hipFuncSetAttribute((const void*)policy_forward_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
#ifdef FPV_EXPERIMENTS
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)policy_forward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
#endif
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int net_first = nets == 2 ? 1 : 0, net_count = nets == 3 ? 2 : 1;
    const size_t lds_bytes = (size_t)(net_count * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float) +
                             (merge.rows ? kMergeLdsBytes : 0);
    PolicyArgs p{image, obs, rms, eps_in, tick, obs_n, action, clipped, value, logp, eps_out,
                 n, seed, id_offset, clip_obs, eps_norm, deterministic, in_dim, act_dim, merge, n_dev,
                 sc.t_idx, sc.d_idx, sc.out, sc.stride, sc.scale};
    const int threads = narrow ? 256 : 512, chunk = threads / 2;
    int64_t blocks = (n + chunk - 1) / chunk;
    if (blocks > max_blocks) blocks = max_blocks;
#ifdef FPV_EXPERIMENTS   // the 256-thread form exists for the critic-beside-the-env-step experiment only
    if (narrow)
        hipLaunchKernelGGL(policy_forward_kernel<256>, dim3((unsigned)blocks), dim3(256), lds_bytes, s, p, net_first, net_count);
    else
#else
    if (narrow) return hipErrorNotSupported;
#endif
        hipLaunchKernelGGL(policy_forward_kernel<512>, dim3((unsigned)blocks), dim3(512), lds_bytes, s, p, net_first, net_count);
    return hipGetLastError();
}

int policy_image_floats() { return kImageFloats; }

}  // namespace fpv
