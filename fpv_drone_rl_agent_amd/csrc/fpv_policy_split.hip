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
#include "fpv_policy_split.h"

namespace fpv {

struct SplitParams {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *wa, *ba, *log_std;
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *wv, *bv;
    int in_dim, act_dim;      // 20 / 4 hover, 12 / 1 yaw: W1's k and the action head's rows beyond them are zero in the image
};

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
            float w = w2[(m * 32 + (lane & 31)) * 128 + kt * 32 + frag_row(s, j, lane >> 5)];
            if (FPV_SPLIT_TANH_FOLD) w *= -2.0f * kTanhPrescale;     // layer 2 consumes r1 = (1 - tanh) / 2 and feeds a prescaled tanh (fpv_mlp.h)
            reinterpret_cast<__bf16*>(base)[o] = piece(w, part);
        } else if (o < kW2Elems + kW1Elems) {   // [m][s][part][lane][j]: input index k = 12 h + 8 s + j (< 12 per half, < 20)
            const int q = o - kW2Elems;
            const int j = q % 8, lane = (q / 8) % 64, part = (q / 512) % 3, s = (q / 1536) % 2, m = q / 3072;
            const int within = 8 * s + j, k = 12 * (lane >> 5) + within;
            float w = (within < 12 && k < p.in_dim) ? w1[(m * 32 + (lane & 31)) * p.in_dim + k] : 0.0f;
            if (FPV_SPLIT_TANH_FOLD) w *= kTanhPrescale;
            reinterpret_cast<__bf16*>(base + kW2Bytes)[q] = piece(w, part);
        } else {                                // biases in accumulator order [layer][m][h][r]
            const int q = o - kW2Elems - kW1Elems, layer = q / 128, z = q % 128, r = z % 16, h = (z / 16) % 2, m = z / 32;
            const int row = m * 32 + acc_row(r, h);
            float bv_ = (layer ? b2 : b1)[row];
            if (FPV_SPLIT_TANH_FOLD) {          // b + sum_k W_k for the layer that consumes r (in double, rounded once), both prescaled
                double b = (double)bv_;
                if (layer)
                    for (int k = 0; k < 128; ++k) b += (double)w2[row * 128 + k];
                bv_ = (float)b * kTanhPrescale;
            }
            reinterpret_cast<float*>(base + kW2Bytes + kW1Bytes)[q] = bv_;
        }
        return;
    }
    const int q = e - 2 * kNetElems;
    if (q >= kHeadFloats) return;
    float v = 0.0f;
    if (q < 512) {                              // Wa[h][kt][r][o]
        const int o = q % 4, r = (q / 4) % 16, kt = (q / 64) % 4, h = q / 256;
        v = o < p.act_dim ? p.wa[o * 128 + kt * 32 + acc_row(r, h)] : 0.0f;
        if (FPV_SPLIT_TANH_FOLD) v *= -2.0f;          // the heads consume r2
    } else if (q < 640) {
        const int z = q - 512, r = z % 16, kt = (z / 16) % 4, h = z / 64;
        v = p.wv[kt * 32 + acc_row(r, h)];
        if (FPV_SPLIT_TANH_FOLD) v *= -2.0f;
    } else if (q < 644) {
        double b = q - 640 < p.act_dim ? (double)p.ba[q - 640] : 0.0;
        if (FPV_SPLIT_TANH_FOLD && q - 640 < p.act_dim)
            for (int k = 0; k < 128; ++k) b += (double)p.wa[(q - 640) * 128 + k];
        v = (float)b;
    } else if (q == 644) {
        double b = (double)p.bv[0];
        if (FPV_SPLIT_TANH_FOLD)
            for (int k = 0; k < 128; ++k) b += (double)p.wv[k];
        v = (float)b;
    }
    else if (q >= 648 && q < 652) v = q - 648 < p.act_dim ? p.log_std[q - 648] : 0.0f;
    else if (q >= 652 && q < 656) v = q - 652 < p.act_dim ? expf(p.log_std[q - 652]) : 0.0f;
    else if (q == 656) {                        // -sum log_std - act_dim * 0.5 ln 2 pi
        if (p.act_dim == 4) v = -(p.log_std[0] + p.log_std[1] + p.log_std[2] + p.log_std[3]) - 2.0f * 1.8378770664093453f;
        else for (int j = 0; j < p.act_dim; ++j) v -= p.log_std[j] + 0.5f * 1.8378770664093453f;
    }
    reinterpret_cast<float*>(img + 2 * (size_t)kNetBytes)[q] = v;
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
    int32_t net_first, net_count;     // which nets: {0, 2} both (pi, then V), {0, 1} pi only, {1, 1} V only
    MergeArgs mg;
};

template <int IN, int ACT>
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
    } else if (threadIdx.x < IN) {
        nrm[threadIdx.x] = p.rms ? (float)p.rms[threadIdx.x] : 0.0f;
        nrm[20 + threadIdx.x] = p.rms ? (float)(1.0 / sqrt(p.rms[IN + threadIdx.x] + (double)p.eps_norm)) : 1.0f;
    }

#pragma unroll 1
    for (int net = p.net_first; net < p.net_first + p.net_count; ++net) {
        __syncthreads();   // everyone is done with the previous image
        split_stage_image<kSplitThreads>(p.image, net, lds_raw);
        __syncthreads();
        const SplitOut out{net == p.net_first ? p.obs_n : nullptr, p.action, p.clipped, p.value, p.logp, p.eps_out};
        const uint64_t tick = p.tick ? p.tick[0] : 0ull;
        for (int64_t base = (int64_t)blockIdx.x * kSplitDrones; base < p.n; base += (int64_t)gridDim.x * kSplitDrones) {
            const int64_t d = base + wave * 32 + col;
            split_net_row<IN, ACT>(net, lds_raw, head, nrm, clip, p.obs, d, d < p.n, lane, out, p.eps_in, tick, p.deterministic != 0, p.seed, p.id_offset);
        }
    }
}

hipError_t launch_policy_split_pack(const float* const* w, unsigned char* image, int in_dim, int act_dim, hipStream_t s) {
    SplitParams p{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], in_dim, act_dim};
    constexpr int elems = 2 * (kW2Bytes / 2 + kW1Bytes / 2 + 256) + kHeadFloats;
    hipLaunchKernelGGL(policy_split_pack_kernel, dim3((elems + 255) / 256), dim3(256), 0, s, p, image);
    return hipGetLastError();
}

hipError_t launch_policy_split_forward(const unsigned char* image, const float* obs, const double* rms, const float* eps_in,
                                       const uint64_t* tick, float* obs_n, float* action, float* clipped, float* value,
                                       float* logp, float* eps_out, int64_t n, uint64_t seed, uint64_t id_offset,
                                       float clip_obs, float eps_norm, int deterministic, int max_blocks, int nets, int yaw_widths,
                                       const MergeArgs& merge, hipStream_t s) {
    static bool attr_tab[64] = {};                    // hipFuncSetAttribute holds for one device: one flag per ordinal
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    bool& attr_set = attr_tab[dev_ & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)policy_split_forward_kernel<kIn, 4>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)policy_split_forward_kernel<kYawObsWords, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kSplitLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    SplitArgs p{image, obs, rms, eps_in, tick, obs_n, action, clipped, value, logp, eps_out,
                n, seed, id_offset, clip_obs, eps_norm, deterministic, (nets & 1) ? 0 : 1, nets == 3 ? 2 : 1, merge};
    int64_t blocks = (n + kSplitDrones - 1) / kSplitDrones;
    if (blocks > max_blocks) blocks = max_blocks;
    if (yaw_widths) hipLaunchKernelGGL((policy_split_forward_kernel<kYawObsWords, 1>), dim3((unsigned)blocks), dim3(kSplitThreads), kSplitLdsBytes, s, p);
    else hipLaunchKernelGGL((policy_split_forward_kernel<kIn, 4>), dim3((unsigned)blocks), dim3(kSplitThreads), kSplitLdsBytes, s, p);
    return hipGetLastError();
}

int policy_split_image_bytes() { return kSplitImageBytes; }

}  // namespace fpv
