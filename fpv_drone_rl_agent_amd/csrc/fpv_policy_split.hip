// fpv_policy_split.hip — the same fused actor-critic forward as fpv_policy.hip, with the two hidden GEMMs on
// bf16 MFMA through a 6-term split that keeps fp32 accuracy:
//   every fp32 operand v is cut into three bf16 pieces v = v1 + v2 + v3 (8 + 8 + 8 significand bits, exact up to
//   the last rounding), and x.w is summed from the six piece products whose order i + j <= 4
//   (x1w1, x1w2, x2w1, x1w3, x3w1, x2w2).  Each piece product is exact in the fp32 accumulator; the dropped
//   terms are below 2^-22 of |x.w|, i.e. at the level of one fp32 rounding — see tests/test_policy_gpu.py, which
//   holds both kernels to the same error against an fp64 reference.
// v_mfma_f32_32x32x16_bf16 retires K = 16 in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64, so six of them
// still cost 2.7x fewer matrix-pipe cycles.  Activations stay in registers between layers exactly as in the fp32
// kernel (accumulator tile -> next layer's B fragment; k order baked into the weight image).  The split weight
// image of one net is 121 KB, so the workgroup makes two passes over its drones: pi (and the sampling epilogue)
// with the pi image resident, then V with the V image.
#include "fpv_mlp.h"

namespace fpv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kIn = 20;
constexpr int kSplitThreads = 512;
constexpr int kSplitDrones = kSplitThreads / 2;   // 32 drones per wave
// per-net image, byte offsets
constexpr int kW2Bytes = 4 * 4 * 2 * 3 * 64 * 16;   // [m][kt][s][part][lane][8 bf16] = 98 304
constexpr int kW1Bytes = 4 * 2 * 3 * 64 * 16;       // [m][s][part][lane][8 bf16]     = 24 576
constexpr int kBiasBytes = 2 * 128 * 4;             // b1p, b2p (fp32, accumulator order)
constexpr int kNetBytes = kW2Bytes + kW1Bytes + kBiasBytes;   // 123 904
constexpr int kHeadFloats = 660;                    // Wa[512] Wv[128] ba[4] bv[1] pad log_std[4] std[4] logp_const
constexpr int kSplitImageBytes = 2 * kNetBytes + kHeadFloats * 4;
constexpr int kSplitLdsBytes = kNetBytes + kHeadFloats * 4 + 40 * 4 + kMergeLdsBytes;   // one net + heads + normaliser + merge exchange

__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k index (inside a 32-row tile) of element j of lane half h at k-step s when an accumulator tile is the B operand
__host__ __device__ constexpr int frag_row(int s, int j, int h) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

struct SplitParams {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
};

__device__ __forceinline__ __bf16 piece(float v, int part) {
    const __bf16 p1 = (__bf16)v;
    if (part == 0) return p1;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    if (part == 1) return p2;
    return (__bf16)(r1 - (float)p2);
}

__global__ void policy_split_pack_kernel(SplitParams p, unsigned char* __restrict__ img) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // one bf16 / fp32 element per thread
    constexpr int kW2Elems = kW2Bytes / 2, kW1Elems = kW1Bytes / 2, kNetElems = kW2Elems + kW1Elems + 256;
    if (e < 2 * kNetElems) {
        const int net = e / kNetElems, o = e % kNetElems;
        const float* w1 = net ? p.vf_w1 : p.pi_w1;
        const float* b1 = net ? p.vf_b1 : p.pi_b1;
        const float* w2 = net ? p.vf_w2 : p.pi_w2;
        const float* b2 = net ? p.vf_b2 : p.pi_b2;
        unsigned char* base = img + (size_t)net * kNetBytes;
        if (o < kW2Elems) {            // [m][kt][s][part][lane][j]
            const int j = o % 8, lane = (o / 8) % 64, part = (o / 512) % 3, s = (o / 1536) % 2, kt = (o / 3072) % 4, m = o / 12288;
            const float w = w2[(m * 32 + (lane & 31)) * 128 + kt * 32 + frag_row(s, j, lane >> 5)];
            reinterpret_cast<__bf16*>(base)[o] = piece(w, part);
        } else if (o < kW2Elems + kW1Elems) {   // [m][s][part][lane][j]: input index k = 12 h + 8 s + j (< 12 per half, < 20)
            const int q = o - kW2Elems;
            const int j = q % 8, lane = (q / 8) % 64, part = (q / 512) % 3, s = (q / 1536) % 2, m = q / 3072;
            const int within = 8 * s + j, k = 12 * (lane >> 5) + within;
            const float w = (within < 12 && k < kIn) ? w1[(m * 32 + (lane & 31)) * kIn + k] : 0.0f;
            reinterpret_cast<__bf16*>(base + kW2Bytes)[q] = piece(w, part);
        } else {                                // biases in accumulator order [layer][m][h][r]
            const int q = o - kW2Elems - kW1Elems, layer = q / 128, z = q % 128, r = z % 16, h = (z / 16) % 2, m = z / 32;
            reinterpret_cast<float*>(base + kW2Bytes + kW1Bytes)[q] = (layer ? b2 : b1)[m * 32 + acc_row(r, h)];
        }
        return;
    }
    const int q = e - 2 * kNetElems;
    if (q >= kHeadFloats) return;
    float v = 0.0f;
    if (q < 512) {                              // Wa[h][kt][r][o]
        const int o = q % 4, r = (q / 4) % 16, kt = (q / 64) % 4, h = q / 256;
        v = p.wa[o * 128 + kt * 32 + acc_row(r, h)];
    } else if (q < 640) {
        const int z = q - 512, r = z % 16, kt = (z / 16) % 4, h = z / 64;
        v = p.wv[kt * 32 + acc_row(r, h)];
    } else if (q < 644) v = p.ba[q - 640];
    else if (q == 644) v = p.bv[0];
    else if (q >= 648 && q < 652) v = p.log_std[q - 648];
    else if (q >= 652 && q < 656) v = expf(p.log_std[q - 652]);
    else if (q == 656) v = -(p.log_std[0] + p.log_std[1] + p.log_std[2] + p.log_std[3]) - 2.0f * 1.8378770664093453f;
    reinterpret_cast<float*>(img + 2 * (size_t)kNetBytes)[q] = v;
}


// three bf16 fragments of eight fp32 values
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        const __bf16 c = (__bf16)(r1 - (float)b);
        p1[j] = a; p2[j] = b; p3[j] = c;
    }
}

// acc += W . X for one k-step of 16 with the six retained piece products, small terms first
__device__ __forceinline__ f32x16 mfma6(const bf16x8* __restrict__ wfrag /* 3 parts, stride 64 fragments */, int lane,
                                        const bf16x8& x1, const bf16x8& x2, const bf16x8& x3, f32x16 acc) {
    const bf16x8 w1 = wfrag[lane], w2 = wfrag[64 + lane], w3 = wfrag[128 + lane];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc, 0, 0, 0);
    return acc;
}

struct SplitArgs {
    const unsigned char* image;
    const float* obs;
    const double* rms;
    const float* eps_in;
    const uint64_t* tick;
    float *obs_n, *action, *clipped, *value, *logp, *eps_out;
    int64_t n;
    uint64_t seed, id_offset;
    float clip_obs, eps_norm;
    int32_t deterministic;
    MergeArgs mg;
};

__global__ __launch_bounds__(kSplitThreads) void policy_split_forward_kernel(SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* head = reinterpret_cast<float*>(lds_raw + kNetBytes);            // kHeadFloats, then mean[20], inv[20]
    float* nrm = head + kHeadFloats;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
    const float clip = p.rms ? p.clip_obs : 3.0e38f;

    // heads + normaliser once
    for (int q = threadIdx.x; q < kHeadFloats; q += kSplitThreads)
        head[q] = reinterpret_cast<const float*>(p.image + 2 * (size_t)kNetBytes)[q];
    if (p.mg.rows) {                            // fold the previous env step's tile sums into the statistics first
        MergeRegs mr;
        merge_request(p.mg, mr);
        merge_prologue(p.mg, mr, p.rms, reinterpret_cast<double*>(nrm + 40), p.eps_norm, nrm, nrm + 20);
    } else if (threadIdx.x < kIn) {
        nrm[threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
        nrm[20 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[kIn + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
    }

#pragma unroll 1
    for (int net = 0; net < 2; ++net) {
        __syncthreads();   // everyone is done with the previous image
        {                  // stage this net's image: all requests in flight before the first LDS write
            const f32x4* src = reinterpret_cast<const f32x4*>(p.image + (size_t)net * kNetBytes);
            f32x4* dst = reinterpret_cast<f32x4*>(lds_raw);
            constexpr int kGran = kNetBytes / 16, kPer = (kGran + kSplitThreads - 1) / kSplitThreads;
            f32x4 stage[kPer];
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * kSplitThreads + threadIdx.x;
                stage[q] = src[g < kGran ? g : 0];
            }
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                const int g = q * kSplitThreads + threadIdx.x;
                if (g < kGran) dst[g] = stage[q];
            }
        }
        __syncthreads();
        const bf16x8* w2img = reinterpret_cast<const bf16x8*>(lds_raw);
        const bf16x8* w1img = reinterpret_cast<const bf16x8*>(lds_raw + kW2Bytes);
        const float* bias = reinterpret_cast<const float*>(lds_raw + kW2Bytes + kW1Bytes);

        for (int64_t base = (int64_t)blockIdx.x * kSplitDrones; base < p.n; base += (int64_t)gridDim.x * kSplitDrones) {
            const int64_t d = base + wave * 32 + col;
            const bool live = d < p.n;
            // observation slice of this lane half (k = 12 h .. 12 h + 11), normalised
            float x[16];
            {
                const f32x4* row = reinterpret_cast<const f32x4*>(p.obs + (live ? d : 0) * kIn);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool have = live && (h == 0 || q < 2);
                    const f32x4 v = have ? row[3 * h + q] : f32x4{0.f, 0.f, 0.f, 0.f};
                    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int k = 12 * h + j;
                    if (k < kIn) x[j] = fminf(fmaxf((x[j] - nrm[k]) * nrm[20 + k], -clip), clip);
                }
                x[12] = x[13] = x[14] = x[15] = 0.0f;
                if (net == 0 && p.obs_n && live) {
                    f32x4* out = reinterpret_cast<f32x4*>(p.obs_n + d * kIn);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (h == 0 || q < 2) out[3 * h + q] = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                }
            }
            // ---- layer 1 (K = 20 padded to 32: two k-steps)
            f32x16 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[m][r] = bias[(m * 2 + h) * 16 + r];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = x[8 * s + j];
                bf16x8 x1, x2, x3;
                split8(v, x1, x2, x3);
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = mfma6(w1img + (m * 2 + s) * 3 * 64, lane, x1, x2, x3, a[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[m][r] = fast_tanh(a[m][r]);
            // ---- layer 2: accumulator tiles of layer 1 are the B operands (registers 8 s .. 8 s + 7 per k-step)
            f32x16 c[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[m][r] = bias[128 + (m * 2 + h) * 16 + r];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = a[kt][8 * s + j];
                    bf16x8 x1, x2, x3;
                    split8(v, x1, x2, x3);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        c[m] = mfma6(w2img + ((m * 4 + kt) * 2 + s) * 3 * 64, lane, x1, x2, x3, c[m]);
                    __builtin_amdgcn_sched_barrier(0);   // keep the next k-step's 12 weight fragments from being hoisted here
                }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) c[m][r] = fast_tanh(c[m][r]);

            // The two lane halves are combined with half_sum() (v_permlane32_swap, VALU only).  The first version went
            // through __shfl_xor -> ds_bpermute_b32 and about 1 row in 3 000 came out with a wrong FIRST head component:
            // the ds_bpermute that read mu[0] issued two instructions after the v_pk_fma_f32 that wrote it and, while
            // the SIMD's partner wave was streaming bf16 MFMAs, picked up the stale register (1 311 bad rows in 3.9 x 10^6
            // with -DFPV_HALFSUM_BPERMUTE, 0 in 2 x 10^7 with the VALU swap; tools/stress_policy.py).  A workgroup
            // barrier here used to hide it by keeping MFMA loops and epilogues apart; it is no longer needed.
            if (net == 0) {
                // ---- action head on the VALU, sampling epilogue
                float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(head + (((h * 4 + kt) * 16 + r) * 4));
                        const float v = c[kt][r];
                        mu[0] += w.x * v; mu[1] += w.y * v; mu[2] += w.z * v; mu[3] += w.w * v;
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) mu[j] = half_sum(mu[j]) + head[640 + j];
                float eps[4] = {0.f, 0.f, 0.f, 0.f};
                if (!p.deterministic) {
                    if (p.eps_in) {
                        const f32x4 e = live ? reinterpret_cast<const f32x4*>(p.eps_in)[d] : f32x4{0.f, 0.f, 0.f, 0.f};
                        eps[0] = e.x; eps[1] = e.y; eps[2] = e.z; eps[3] = e.w;
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
                float act[4], lp = head[656];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    act[j] = mu[j] + head[652 + j] * eps[j];
                    lp -= 0.5f * eps[j] * eps[j];
                }
                if (live) {
                    if (h == 0) {
                        if (p.action) reinterpret_cast<f32x4*>(p.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
                        if (p.clipped)
                            reinterpret_cast<f32x4*>(p.clipped)[d] =
                                f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f),
                                      fminf(fmaxf(act[2], -1.f), 1.f), fminf(fmaxf(act[3], -1.f), 1.f)};
                    } else {
                        if (p.logp) p.logp[d] = lp;
                        if (p.eps_out) reinterpret_cast<f32x4*>(p.eps_out)[d] = f32x4{eps[0], eps[1], eps[2], eps[3]};
                    }
                }
            } else {
                float v = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v += head[512 + (h * 4 + kt) * 16 + r] * c[kt][r];
                v = half_sum(v);
                if (live && h == 1) p.value[d] = v + head[644];
            }
        }
    }
}

hipError_t launch_policy_split_pack(const float* const* w, unsigned char* image, hipStream_t s) {
    SplitParams p{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12]};
    constexpr int elems = 2 * (kW2Bytes / 2 + kW1Bytes / 2 + 256) + kHeadFloats;
    hipLaunchKernelGGL(policy_split_pack_kernel, dim3((elems + 255) / 256), dim3(256), 0, s, p, image);
    return hipGetLastError();
}

hipError_t launch_policy_split_forward(const unsigned char* image, const float* obs, const double* rms, const float* eps_in,
                                       const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                       float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                       float clip_obs, float eps_norm, int deterministic, int max_blocks,
                                       const MergeArgs& merge, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)policy_split_forward_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    SplitArgs p{image, obs, rms, eps_in, tick, obs_n, action, clipped, value, logp, eps_out,
                n, seed, id_offset, clip_obs, eps_norm, deterministic, merge};
    int64_t blocks = (n + kSplitDrones - 1) / kSplitDrones;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(policy_split_forward_kernel, dim3((unsigned)blocks), dim3(kSplitThreads), kSplitLdsBytes, s, p);
    return hipGetLastError();
}

int policy_split_image_bytes() { return kSplitImageBytes; }

}  // namespace fpv
