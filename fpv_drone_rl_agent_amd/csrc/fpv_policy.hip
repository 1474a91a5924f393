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

namespace fpv {

constexpr int kHid = 128, kIn = 20, kInPad = 24, kAct = 4;
constexpr int kNetFloats = 4 * 12 * 64 + 128 + 4 * 4 * 4 * 64 * 4 + 128;  // 19712
constexpr int kOffW1 = 0, kOffB1 = 3072, kOffW2 = 3200, kOffB2 = 19584;
constexpr int kHeadBase = 2 * kNetFloats;                                   // 39424
constexpr int kOffWa = kHeadBase, kOffWv = kHeadBase + 512, kOffBa = kHeadBase + 640, kOffBv = kHeadBase + 644,
              kOffLogStd = kHeadBase + 648, kOffStd = kHeadBase + 652, kOffLogpConst = kHeadBase + 656;
constexpr int kImageFloats = kHeadBase + 660;                               // packed image
// LDS: the staged nets, then the 660 head floats, then the normaliser mean[20] / inv_std[20] (160 496 B for both nets)

// ---------------------------------------------------------------- weight image
// nn.Linear layouts: w1 [128][20], w2 [128][128], wa [4][128], wv [1][128] (row = output).
struct PolicyParams {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
    int in_dim, act_dim;   // 20 / 4 hover (hover.py:59-70), 12 / 1 yaw (yaw.py:37-45); in_dim % 4 == 0, <= 24; act_dim <= 4
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
        if (o < kOffB1) {                       // W1p[m][s][lane] = W1[m*32 + i][s + 12 h]
            const int lane = o % 64, s = (o / 64) % 12, m = o / (64 * 12);
            const int k = s + 12 * (lane >> 5);
            v = k < p.in_dim ? w1[(m * 32 + (lane & 31)) * p.in_dim + k] : 0.0f;
        } else if (o < kOffW2) {                // b1p[m][h][r]
            const int q = o - kOffB1, r = q % 16, h = (q / 16) % 2, m = q / 32;
            v = b1[m * 32 + tile_row(r, h)];
        } else if (o < kOffB2) {                // W2p[m][kt][r4][lane][rr] = W2[m*32 + i][kt*32 + row(4 r4 + rr, h)]
            const int q = o - kOffW2, rr = q % 4, lane = (q / 4) % 64, r4 = (q / 256) % 4, kt = (q / 1024) % 4, m = q / 4096;
            v = w2[(m * 32 + (lane & 31)) * kHid + kt * 32 + tile_row(4 * r4 + rr, lane >> 5)];
        } else {
            const int q = o - kOffB2, r = q % 16, h = (q / 16) % 2, m = q / 32;
            v = b2[m * 32 + tile_row(r, h)];
        }
    } else {
        const int q = e - kHeadBase;
        if (q < 512) {                          // Wa[h][kt][r][o]
            const int o = q % 4, r = (q / 4) % 16, kt = (q / 64) % 4, h = q / 256;
            v = o < p.act_dim ? p.wa[o * kHid + kt * 32 + tile_row(r, h)] : 0.0f;
        } else if (q < 640) {                   // Wv[h][kt][r]
            const int z = q - 512, r = z % 16, kt = (z / 16) % 4, h = z / 64;
            v = p.wv[kt * 32 + tile_row(r, h)];
        } else if (q < 644) v = q - 640 < p.act_dim ? p.ba[q - 640] : 0.0f;
        else if (q == 644) v = p.bv[0];
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
    if (p.n_dev) {
        const int64_t nd = *p.n_dev;
        rows = nd < rows ? (nd > 0 ? nd : 0) : rows;
    }
    for (int64_t base = (int64_t)blockIdx.x * kChunk; base < rows; base += (int64_t)gridDim.x * kChunk) {
        const int64_t d = base + wave * 32 + col;   // this lane's drone (both halves of the wave share it)
        const bool live = d < rows;
        // ---- observation: half 0 holds k = 0..11, half 1 holds k = 12..23 (20..23 are zero padding)
        float x[12];
        {
            const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + (live ? d : 0) * p.in_dim);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool have = live && 4 * (3 * h + q) < p.in_dim;
                const f32x4 v = have ? row[3 * h + q] : f32x4{0.f, 0.f, 0.f, 0.f};
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int k = 12 * h + j;
                if (k < p.in_dim) x[j] = fminf(fmaxf((x[j] - hb[660 + k]) * hb[680 + k], -clip), clip);
            }
            if (p.obs_n && live) {
                f32x4* out = reinterpret_cast<f32x4*>(p.obs_n + d * p.in_dim);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (4 * (3 * h + q) < p.in_dim) out[3 * h + q] = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
            }
        }

        FPV_STAMP(3);
        float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // 4 action means + value (partial sums of this lane half)
#pragma unroll 1
        for (int sl = 0; sl < net_count; ++sl) {
            const int net = net_first + sl;
            const float* nb = lds + sl * kNetFloats;
            // ---- layer 1: 20 (padded 24) -> 128
            Hidden a;
            bias_init(a, nb + kOffB1, h);
#pragma unroll
            for (int s = 0; s < 12; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(nb[kOffW1 + (m * 12 + s) * 64 + lane], x[s], a.t[m], 0, 0, 0);
            FPV_STAMP(4 + 5 * sl);
            tanh_all(a);
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
            tanh_all(c);
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
        if (do_v && live && h == 1) p.value[d] = head[4] + hb[644];
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

// ---------------------------------------------------------------- pi + V in one software pipeline
// The rollout's launch (both nets, 512 threads).  Phase stamps of the kernel above (profiles/r02_policy_phase_stamps.md)
// showed where a 65 536-drone launch loses the matrix pipe: (i) ~10 k cycles pass before the first product — 160 KB of
// weight image through registers into LDS, then the observation load; (ii) the SIMD's matrix pipe is arbitrated
// oldest-wave-first, so the older wave of a SIMD runs through its layers at full rate while the younger one only fills
// its gaps and then runs the second half of the launch alone — and alone, every tanh / head / sampling phase of a wave
// leaves the pipe idle.  This kernel removes both:
//  * the image comes in by LDS-DMA (global_load_lds_dwordx4, no registers), issued in consumption order
//    {pi.W1 | pi.W2 + heads | V.W1 | V.W2}; a wave waits (counted vmcnt + raw s_barrier) only for what its next phase
//    reads, so the first product issues once 13 KB have landed and the rest arrives under the layer-1 products;
//  * every layer is computed one 32-row output tile at a time (a dependent chain of products on one accumulator runs
//    at the full rate), and the vector work of a finished tile — tanh, head terms, and for the action head the
//    sampling — sits between the products of the next tile, also across the layer and net boundaries.  A wave is then
//    matrix-dense on its own, whichever way the SIMD arbitrates; only the tanh + head of the very last tile is exposed.
//    Two layers' tiles are live at a time (<= 190 registers, no scratch).
// Arithmetic per output element is the same fma chain in the same order as the kernel above: results are bit-identical.
__device__ __forceinline__ void load_obs_rows(const PolicyArgs& p, int64_t d, bool live, int h, float (&x)[12]) {
    const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + (live ? d : 0) * p.in_dim);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const bool have = live && 4 * (3 * h + q) < p.in_dim;
        const f32x4 v = have ? row[3 * h + q] : f32x4{0.f, 0.f, 0.f, 0.f};
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
}

// One 1-KiB piece of the image by LDS-DMA.  Inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write
// and drains vmcnt(0) in front of the next ds_read, which would serialise the staging with the first layer; an asm
// statement is outside its bookkeeping (cdna_hip_programming.md §5.7), the counted waits below are ours.  M0 (the
// wave-uniform LDS destination) is written and restored inside the statement.
__device__ __forceinline__ void stage_piece(const float* __restrict__ image, unsigned lds_base, int piece, int lane, int image_bytes) {
    const int byte = piece * 1024 + lane * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)piece * 1024u);
    const float* src = image + piece * 256 + lane * 4;
    unsigned keep;
    if (byte < image_bytes)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

#define FPV_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FPV_PIN() __builtin_amdgcn_sched_barrier(0)

// LDS reads of the pipeline go through explicit address-space-3 pointers whose base passes through an empty asm: hipcc
// otherwise turns the few hundred constant LDS addresses of the unrolled pipeline (the 160 KB image is beyond one
// 16-bit ds offset) into as many loop-invariant base registers, hoists them out of the chunk loop and spills them.
// Behind an opaque base every access is base + immediate.
typedef __attribute__((address_space(3))) const float lfloat;
typedef __attribute__((address_space(3))) const f32x4 lf32x4;
__device__ __forceinline__ lfloat* lds_base(unsigned byte_addr) {
    asm volatile("" : "+v"(byte_addr));
    return (lfloat*)(size_t)byte_addr;
}
__device__ __forceinline__ f32x4 ld4(lfloat* b, int float_off) { return *(lf32x4*)(b + float_off); }

// ACT4: act_dim == 4 (hover) — the action rows leave as 16-byte stores.  The kernel samples with the in-kernel Philox
// stream; eps_in / deterministic / n_dev / single-net launches stay with the kernel above (not on the rollout's path).
template <bool ACT4>
__global__ __launch_bounds__(512) void policy_forward_dual_kernel(PolicyArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* hb = lds_f + 2 * kNetFloats;              // heads [660], then normaliser mean[20], inv[20]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_f;   // LDS byte address of the image
    const unsigned hb0 = lds0 + 2 * kNetFloats * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const float clip = p.rms ? p.clip_obs : 3.0e38f;
    constexpr int kChunk = 256;
    FPV_STAMP(0);

    const int64_t rows = p.n;
    const int64_t base0 = (int64_t)blockIdx.x * kChunk;
    if (base0 >= rows) return;                       // uniform: nothing to stage for
    const uint64_t tick = p.tick ? p.tick[0] : 0ull;

    // ---- the wave's first vector-memory operations: normaliser statistics (wave 0) and the first chunk's observation
    if ((int)threadIdx.x < p.in_dim) {
        hb[660 + threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
        hb[680 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[p.in_dim + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
    }
    float x[12];
    // first chunk: asm loads, so that hipcc's own wait for them does not become vmcnt(0) behind the LDS-DMA it cannot see
    f32x4 xq0, xq1, xq2;
    {
        const int64_t d = base0 + wave * 32 + col;
        const float* row = p.obs + (d < rows ? d : 0) * p.in_dim;
        const float* a0 = row + ((4 * (3 * h + 0) < p.in_dim) ? 4 * (3 * h + 0) : 0);
        const float* a1 = row + ((4 * (3 * h + 1) < p.in_dim) ? 4 * (3 * h + 1) : 0);
        const float* a2 = row + ((4 * (3 * h + 2) < p.in_dim) ? 4 * (3 * h + 2) : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq0) : "v"(a0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq1) : "v"(a1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq2) : "v"(a2) : "memory");
    }
    // ---- weight image by LDS-DMA, 21 one-KiB pieces per wave, in consumption order (a slot past its group's end
    // re-loads the group's last piece: same bytes to the same place)
    {
        constexpr int kBytes = kImageFloats * 4;     // 160 336: piece 156 is partial (exec-masked lanes write nothing)
#pragma unroll
        for (int it = 0; it < 2; ++it) {             // G1: pieces 0..12 = pi.W1, pi.b1
            const int j = it * 8 + wave;
            stage_piece(p.image, lds0, j < 13 ? j : 12, lane, kBytes);
        }
#pragma unroll
        for (int it = 0; it < 9; ++it) {             // G2: pieces 13..76 = pi.W2, pi.b2; 154..156 = heads
            const int j = it * 8 + wave;
            stage_piece(p.image, lds0, j < 64 ? 13 + j : (j < 67 ? 154 + (j - 64) : 156), lane, kBytes);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {             // G3: pieces 77..89 = V.W1, V.b1
            const int j = it * 8 + wave;
            stage_piece(p.image, lds0, j < 13 ? 77 + j : 89, lane, kBytes);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) stage_piece(p.image, lds0, 90 + it * 8 + wave, lane, kBytes);   // G4: V.W2, V.b2
    }
    FPV_STAMP(1);
    {
        // observation + G1 landed (G2, G3, G4: 19 pieces still in flight); lgkmcnt(0): the normaliser's LDS writes
        asm volatile("s_waitcnt vmcnt(19) lgkmcnt(0)" : "+v"(xq0), "+v"(xq1), "+v"(xq2) :: "memory");
        FPV_PIN();
        __builtin_amdgcn_s_barrier();
        const bool live = base0 + wave * 32 + col < rows;
        const bool h0 = live && 4 * (3 * h + 0) < p.in_dim, h1 = live && 4 * (3 * h + 1) < p.in_dim,
                   h2 = live && 4 * (3 * h + 2) < p.in_dim;
        x[0] = h0 ? xq0.x : 0.f; x[1] = h0 ? xq0.y : 0.f; x[2] = h0 ? xq0.z : 0.f; x[3] = h0 ? xq0.w : 0.f;
        x[4] = h1 ? xq1.x : 0.f; x[5] = h1 ? xq1.y : 0.f; x[6] = h1 ? xq1.z : 0.f; x[7] = h1 ? xq1.w : 0.f;
        x[8] = h2 ? xq2.x : 0.f; x[9] = h2 ? xq2.y : 0.f; x[10] = h2 ? xq2.z : 0.f; x[11] = h2 ? xq2.w : 0.f;
    }

    bool first = true;
    for (int64_t base = base0; base < rows; base += (int64_t)gridDim.x * kChunk) {
        const int64_t d = base + wave * 32 + col;   // this lane's drone (both halves of the wave share it)
        const bool live = d < rows;
        // lane-dependent address terms are re-derived inside the loop from laundered values: loop-invariant LDS addresses
        // would be hoisted out of the chunk loop, a register each
        int ln = lane, hh = h;
        asm volatile("" : "+v"(ln), "+v"(hh));
        if (!first) {                               // (the first chunk's rows came in ahead of the weight image)
            load_obs_rows(p, d, live, h, x);
            // the wait for these loads belongs inside this branch: at the join hipcc would place a vmcnt(0) that the
            // first chunk pays with its LDS-DMA still in flight
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                              "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
        }
        FPV_STAMP(2);
        // ---- normalise: half 0 holds k = 0..11, half 1 holds k = 12..23 (20..23 are zero padding)
        {
            lfloat* nrm = lds_base(hb0 + 660 * 4 + 48 * hh);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 mv = ld4(nrm, 4 * q), iv = ld4(nrm, 20 + 4 * q);
                const float mean[4] = {mv.x, mv.y, mv.z, mv.w}, inv[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float nv = fminf(fmaxf((x[4 * q + j] - mean[j]) * inv[j], -clip), clip);
                    x[4 * q + j] = 12 * h + 4 * q + j < p.in_dim ? nv : 0.0f;     // (padding rows of W1 are zero)
                }
            }
        }
        if (p.obs_n && live) {
            f32x4* out = reinterpret_cast<f32x4*>(p.obs_n + d * p.in_dim);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (4 * (3 * h + q) < p.in_dim) out[3 * h + q] = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
        }
        FPV_PIN();
        FPV_STAMP(3);

        float head[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // 4 action means + value (partial sums of this lane half)
        float eps[4] = {0.f, 0.f, 0.f, 0.f};
        f32x16 pend = {};                           // pi's last layer-2 tile, finished under V's first layer-1 tile

        auto tanh_rows = [&](f32x16& t, int r0, int r1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r >= r0 && r < r1) t[r] = fast_tanh(t[r]);
        };
        auto bias_tile = [&](f32x16& t, lfloat* b) {                   // b -> 16 floats
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = ld4(b, 4 * q);
                t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
            }
        };
        // head terms of one row of a finished layer-2 tile: the same instructions for both nets (no branch between the
        // products) — s_pi / s_v are 1 / 0 for pi and 0 / 1 for V, so the other net's term adds an exact zero
        auto head_row = [&](float v, f32x4 wh, float wvr, float s_pi, float s_v) {
            const float t = fast_tanh(v), tp = t * s_pi, tv = t * s_v;
            head[0] += wh.x * tp; head[1] += wh.y * tp; head[2] += wh.z * tp; head[3] += wh.w * tp;
            head[4] += wvr * tv;
        };

        // (the two nets as a real loop: fully unrolled, hipcc's allocator spills ~250 registers of this body; as a loop
        // it needs ~170 and none — the net index then lives in an SGPR)
#pragma unroll 1
        for (int net = 0; net < 2; ++net) {
            const unsigned nb0 = lds0 + net * (kNetFloats * 4);
            const float s_pi = net == 0 ? 1.0f : 0.0f, s_v = 1.0f - s_pi;
            asm volatile("" : "+v"(ln), "+v"(hh));
            lfloat* wa = lds_base(hb0 + hh * 1024);                    // Wa[h][kt][r][o]
            lfloat* wv = lds_base(hb0 + 512 * 4 + hh * 256);           // Wv[h][kt][r]
            // ---- layer 1, tile by tile: 12 chained products per tile; the previous tile's tanh rides between them
            // (under V's first tile: pi's last layer-2 tile — pend is zero with s_pi' = 0 in the pi pass)
            Hidden a;
            lfloat* w1b = lds_base(nb0 + kOffW1 * 4 + ln * 4);         // W1p[m][s][lane]
            float w1[12];
#pragma unroll
            for (int s = 0; s < 12; ++s) w1[s] = w1b[s * 64];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bias_tile(a.t[m], lds_base(nb0 + kOffB1 * 4 + (m * 2) * 64 + hh * 64));
                float w1n[12];
#pragma unroll
                for (int s = 0; s < 12; ++s) w1n[s] = m < 3 ? w1b[((m + 1) * 12 + s) * 64] : 0.0f;   // next tile's fragments
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 wh[4];
                    float wvr[4];
                    if (m == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { wh[j] = ld4(wa, (3 * 16 + 4 * q + j) * 4); wvr[j] = 0.0f; }
                    }
#pragma unroll
                    for (int s = 3 * q; s < 3 * q + 3; ++s)
                        a.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[s], x[s], a.t[m], 0, 0, 0);
                    // after every third product: a quarter of the pending vector work
                    if (m > 0) tanh_rows(a.t[m - 1], 4 * q, 4 * q + 4);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) head_row(pend[4 * q + j], wh[j], wvr[j], s_v, 0.0f);
                    }
                    FPV_PIN();
                }
#pragma unroll
                for (int s = 0; s < 12; ++s) w1[s] = w1n[s];
            }
            if (net == 0) FPV_STAMP(4); else FPV_STAMP(6);
            if (first) {
                if (net == 0) FPV_WAIT_VM(10);      // G2 landed
                else FPV_WAIT_VM(0);                // G4 landed
                __builtin_amdgcn_s_barrier();
            }
            // ---- layer 2, tile by tile: 64 chained products per tile; the layer-1 tiles are the B operands as they stand.
            // Per four products: the next weight fragment and the head weights of the pending row are requested first,
            // then the products, then the vector work of one row of the previous tile
            Hidden c;
            lfloat* w2b = lds_base(nb0 + kOffW2 * 4 + ln * 16);        // W2p[m][kt][r4][lane][rr]
            f32x4 w = ld4(w2b, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bias_tile(c.t[m], lds_base(nb0 + kOffB2 * 4 + (m * 2) * 64 + hh * 64));
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = kt * 4 + r4, nxt = (m * 4 + kt) * 4 + r4 + 1;
                        const f32x4 wn = nxt < 64 ? ld4(w2b, nxt * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 wh = {0.f, 0.f, 0.f, 0.f};
                        float wvr = 0.0f;
                        if (m > 0) { wh = ld4(wa, ((m - 1) * 16 + r) * 4); wvr = wv[(m - 1) * 16 + r]; }
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            c.t[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[rr], a.t[kt][4 * r4 + rr], c.t[m], 0, 0, 0);
                        if (m == 0) {
                            if (kt < 2) tanh_rows(a.t[3], 2 * r, 2 * r + 2);           // done before kt = 3 reads it
                        } else {
                            head_row(c.t[m - 1][r], wh, wvr, s_pi, s_v);
                        }
                        w = wn;
                        FPV_PIN();
                    }
                    if (m == 1 && kt == 0 && net == 1) {
                        // Philox4x32-10 keyed like the env noise (stream tag 'POL1') + Box-Muller
                        uint32_t xr[4];
                        unsigned did = (unsigned)d;     // laundered: keeps the draw here, under the products, instead of
                        asm volatile("" : "+v"(did));   // being hoisted to the top of the chunk as net-loop-invariant
                        philox4x32_10((uint32_t)p.id_offset + did, (uint32_t)tick, (uint32_t)(tick >> 32), 0x504F4C31u,
                                      (uint32_t)p.seed, (uint32_t)(p.seed >> 32), xr);
                        constexpr float k24 = 5.9604644775390625e-8f, m2ln2 = -1.3862943611198906f;
                        const float u0 = ((float)(xr[0] >> 8) + 0.5f) * k24, u1 = ((float)(xr[1] >> 8) + 0.5f) * k24;
                        const float u2 = ((float)(xr[2] >> 8) + 0.5f) * k24, u3 = ((float)(xr[3] >> 8) + 0.5f) * k24;
                        const float ra = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u0)), rb = __builtin_amdgcn_sqrtf(m2ln2 * __log2f(u2));
                        eps[0] = ra * __builtin_amdgcn_cosf(u1); eps[1] = ra * __builtin_amdgcn_sinf(u1);
                        eps[2] = rb * __builtin_amdgcn_cosf(u3); eps[3] = rb * __builtin_amdgcn_sinf(u3);
                        FPV_PIN();
                    }
                }
            }
            if (net == 0) {
                pend = c.t[3];
                FPV_STAMP(5);
                if (first) {
                    FPV_WAIT_VM(8);                 // G3 landed
                    __builtin_amdgcn_s_barrier();
                }
            } else {
                FPV_STAMP(7);
                // ---- action: combine the two lane halves, sample, log-prob, clip, store (pi's head was complete after
                // V's first layer-1 tile; the stores go out ahead of the exposed tail)
                {
                    lfloat* hs = lds_base(hb0 + 640 * 4);              // ba[4], bv, -, -, -, log_std[4], std[4], logp const
                    const float mu[4] = {half_sum(head[0]) + hs[0], half_sum(head[1]) + hs[1], half_sum(head[2]) + hs[2],
                                         half_sum(head[3]) + hs[3]};
                    float act[4], lp = hs[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        act[j] = mu[j] + hs[12 + j] * eps[j];
                        if (j < p.act_dim) lp -= 0.5f * eps[j] * eps[j];
                    }
                    if (live) {
                        if (h == 0) {
                            if constexpr (ACT4) {
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
                                if constexpr (ACT4) reinterpret_cast<f32x4*>(p.eps_out)[d] = f32x4{eps[0], eps[1], eps[2], eps[3]};
                                else for (int j = 0; j < p.act_dim; ++j) p.eps_out[d * p.act_dim + j] = eps[j];
                            }
                        }
                    }
                    // ---- the exposed tail: V's last layer-2 tile
#pragma unroll
                    for (int r = 0; r < 16; ++r) head[4] += wv[3 * 16 + r] * fast_tanh(c.t[3][r]);
                    head[4] = half_sum(head[4]);
                    if (live && h == 1) p.value[d] = head[4] + hs[4];
                }
            }
        }
        first = false;
        FPV_STAMP(8);
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
                                 int in_dim, int act_dim, const MergeArgs& merge, const int32_t* n_dev, int legacy, hipStream_t s) {
    // nets: 3 = pi + V (one launch), 1 = pi only (with the sampling epilogue), 2 = V only.
    // narrow: 256-thread workgroups (one wave per SIMD) so that other kernels can co-reside.
    static bool attr_set = false;
    if (!attr_set) {
        const int full = (int)((2 * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float)) + kMergeLdsBytes;
        hipError_t e = hipFuncSetAttribute((const void*)policy_forward_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)policy_forward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int net_first = nets == 2 ? 1 : 0, net_count = nets == 3 ? 2 : 1;
    if (!legacy && nets == 3 && !narrow && !merge.rows && value && !eps_in && !deterministic && !n_dev) {
        // the rollout's launch: both nets in one software pipeline, in-kernel Philox sampling
        static bool dual_attr = false;
        const int full = (int)((kImageFloats + 40) * sizeof(float));
        if (!dual_attr) {
            hipError_t e = hipFuncSetAttribute((const void*)policy_forward_dual_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)policy_forward_dual_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, full);
            if (e != hipSuccess) return e;
            dual_attr = true;
        }
        PolicyArgs pd{image, obs, rms, eps_in, tick, obs_n, action, clipped, value, logp, eps_out,
                      n, seed, id_offset, clip_obs, eps_norm, deterministic, in_dim, act_dim, merge, n_dev};
        int64_t blocks = (n + 255) / 256;
        if (blocks > max_blocks) blocks = max_blocks;
        if (act_dim == 4) hipLaunchKernelGGL(policy_forward_dual_kernel<true>, dim3((unsigned)blocks), dim3(512), full, s, pd);
        else hipLaunchKernelGGL(policy_forward_dual_kernel<false>, dim3((unsigned)blocks), dim3(512), full, s, pd);
        return hipGetLastError();
    }
    const size_t lds_bytes = (size_t)(net_count * kNetFloats + (kImageFloats - kHeadBase) + 40) * sizeof(float) +
                             (merge.rows ? kMergeLdsBytes : 0);
    PolicyArgs p{image, obs, rms, eps_in, tick, obs_n, action, clipped, value, logp, eps_out,
                 n, seed, id_offset, clip_obs, eps_norm, deterministic, in_dim, act_dim, merge, n_dev};
    const int threads = narrow ? 256 : 512, chunk = threads / 2;
    int64_t blocks = (n + chunk - 1) / chunk;
    if (blocks > max_blocks) blocks = max_blocks;
    if (narrow)
        hipLaunchKernelGGL(policy_forward_kernel<256>, dim3((unsigned)blocks), dim3(256), lds_bytes, s, p, net_first, net_count);
    else
        hipLaunchKernelGGL(policy_forward_kernel<512>, dim3((unsigned)blocks), dim3(512), lds_bytes, s, p, net_first, net_count);
    return hipGetLastError();
}

int policy_image_floats() { return kImageFloats; }

}  // namespace fpv
