// fpv_policy_split.h — the 6-term bf16 split of the actor-critic forward (fpv_policy_split.hip) as device code shared by the
// stand-alone policy kernel and the one-kernel rollout step (fpv_rollstep.hip: rollout_step_split_kernel), so that both run the
// same instructions on a row.  See fpv_policy_split.hip for the arithmetic and the image layout.
#pragma once
#include "fpv_mlp.h"
#include "fpv_policy_image.h"   // kIn

namespace fpv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


// FPV_SPLIT_TANH_FOLD (round 5): what fpv_mlp.h's FPV_TANH_FOLD does for the exact-fp32 image, for the split image — the hidden layers'
// weights and biases carry tanh's 2 log2 e, a hidden layer hands r = 1 / (1 + e^{2x}) = (1 - tanh x) / 2 to its consumer, whose image
// holds -2 W (cut into pieces AFTER the scaling: one rounding of the weight, as in the fp32 image) and b + sum_k W_k (summed in
// double, rounded once): a tanh is exp2, add, rcp — three vector instructions instead of five, 512 fewer per wave.
#ifndef FPV_SPLIT_HEAD_PACKED
#define FPV_SPLIT_HEAD_PACKED 1       // the heads' multiply-adds as v_pk_fma_f32 pairs (0: one v_fmac_f32 per output and hidden unit; A/B switch)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef FPV_SPLIT_TANH_FOLD
#define FPV_SPLIT_TANH_FOLD 1
#endif
__device__ __forceinline__ float split_carrier(float z) {
#if FPV_SPLIT_TANH_FOLD
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
#else
    return fast_tanh(z);
#endif
}

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

__device__ __forceinline__ __bf16 piece(float v, int part) {
    const __bf16 p1 = (__bf16)v;
    if (part == 0) return p1;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    if (part == 1) return p2;
    return (__bf16)(r1 - (float)p2);
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

// The same for FOUR accumulator tiles (the four 32-row output tiles of a layer's k-step), piece by piece across the tiles: two
// products into one accumulator are then four products apart — a 32 x 32 x 16 product holds its accumulator for 16 passes, so six
// products in a row into the same tile (mfma6 tile after tile) wait for each other.  The order of the six terms per tile is
// mfma6's, so the sums are the same bits.
// A scheduling fence behind every k-step keeps hipcc from hoisting the next k-steps' weight fragments over this one's products.
// The stand-alone kernel (fpv_policy_split.hip: a chunk loop around the nets) needs it — without it 571 registers spill to scratch
// —; the one-kernel rollout step (fpv_rollstep.hip: one chunk, no loop) is 1.4 us faster in its policy phase without (no spill:
// 226 registers) and defines the macro empty before it includes this header.  A fence moves no arithmetic: same bits.
#ifndef FPV_SPLIT_FENCE
#define FPV_SPLIT_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef FPV_SPLIT_MFMA_TILEWISE
#define FPV_SPLIT_MFMA_TILEWISE 0     // 1: mfma6 tile after tile (round 4's order; A/B switch)
#endif
__device__ __forceinline__ void mfma6x4(const bf16x8* __restrict__ wfrag /* tile m at wfrag + m * tile_stride */, int tile_stride, int lane,
                                        const bf16x8& x1, const bf16x8& x2, const bf16x8& x3, f32x16 (&acc)[4]) {
#if FPV_SPLIT_MFMA_TILEWISE
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = mfma6(wfrag + m * tile_stride, lane, x1, x2, x3, acc[m]);
#else
    bf16x8 w1[4], w2[4], w3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        w1[m] = wfrag[m * tile_stride + lane]; w2[m] = wfrag[m * tile_stride + 64 + lane]; w3[m] = wfrag[m * tile_stride + 128 + lane];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[m], x1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[m], x3, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[m], x2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[m], x1, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[m], x2, acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[m], x1, acc[m], 0, 0, 0);
#endif
}


// FPV_SPLIT_HAND (round 5, second session): layer 2 as a hand-placed stream.  An MFMA 32x32x16 occupies the matrix pipe for 32 cycles
// and the SIMD's vector issue for 8 of them (MI355X_MICROARCH.md, per-instruction constants): about 24 cycles of vector issue behind
// every product are free IF vector instructions stand there in program order.  hipcc emits a k-step's 24 products back to back and
// the 52 instructions of the next cut as one block, so with two waves per SIMD in step the phase costs the SUM of the two.  Here the
// order is written out and pinned (scheduling barriers): behind every second product half a pair's cut of the NEXT k-step (5 - 6
// instructions), behind the products of the last two groups one tanh of layer 1's tiles 2 and 3 (needed from k-step 4 on), and a
// group's weight fragments are requested for the next k-step into the registers the group has just finished with.  The products
// per accumulator tile and the cut's arithmetic are those of mfma6x4 / split8: the same bits.
#ifndef FPV_SPLIT_HAND
#define FPV_SPLIT_HAND 0
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t split_pk2(float a, float b) {        // v_cvt_pk_bf16_f32: {bf16(a), bf16(b)}, round to nearest even
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ void split_cut_a(float v0, float v1, uint32_t& p1, float& r0, float& r1) {
    p1 = split_pk2(v0, v1);
    r0 = v0 - __uint_as_float(p1 << 16);
    r1 = v1 - __uint_as_float(p1 & 0xffff0000u);
}
__device__ __forceinline__ void split_cut_b(float r0, float r1, uint32_t& p2, uint32_t& p3) {
    p2 = split_pk2(r0, r1);
    p3 = split_pk2(r0 - __uint_as_float(p2 << 16), r1 - __uint_as_float(p2 & 0xffff0000u));
}
#define FPV_PIN() __builtin_amdgcn_sched_barrier(0)

// Which outputs a pass over one net writes for its row (each pointer may be null)
struct SplitOut {
    float *obs_n, *action, *clipped, *value, *logp, *eps_out;
};

// One net (0: pi with the sampling epilogue, 1: V) for this lane's row d, in two parts so that a caller can put work between them
// (the one-kernel rollout step requests the next net's image there): split_net_hidden — observation, both hidden layers, the net's
// split image in LDS at `img`, the normaliser (mean[20], 1 / sigma[20]) at `nrm`; returns tanh of layer 2 as four accumulator tiles —
// and split_net_head — the head block at `head`: action head + sampling epilogue (net 0; returns the clipped action) or the value.
struct SplitHidden {
    f32x16 c[4];
};

// (layer 1 reads the image's W1 + bias block only, layer 2 W2 and the bias block: a caller may run layer 1 while W2 is on its way)
// IN: observation words (20 hover, 12 yaw): lane half h supplies k = 12 h + j, j < 12, as far as k < IN (the image's W1 holds zeros beyond)
template <int IN = kIn>
__device__ __forceinline__ void split_net_layer1(const unsigned char* img, const float* nrm, float clip, const float* __restrict__ obs,
                                                 int64_t d, bool live, int lane, float* obs_n_out, f32x16 (&a)[4]) {
    static_assert(IN % 4 == 0 && IN <= 24, "observation rows are whole 16-byte quads");
    const int h = lane >> 5;
    const bf16x8* w1img = reinterpret_cast<const bf16x8*>(img + kW2Bytes);
    const float* bias = reinterpret_cast<const float*>(img + kW2Bytes + kW1Bytes);
    // observation slice of this lane half (k = 12 h .. 12 h + 11), normalised
    float x[16];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(obs + (live ? d : 0) * IN);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const bool have = live && 12 * h + 4 * q < IN;
            const f32x4 v = have ? row[3 * h + q] : f32x4{0.f, 0.f, 0.f, 0.f};
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int k = 12 * h + j;
            if (k < IN) x[j] = fminf(fmaxf((x[j] - nrm[k]) * nrm[20 + k], -clip), clip);
        }
        x[12] = x[13] = x[14] = x[15] = 0.0f;
        if (obs_n_out && live) {                  // (the caller hands obs_n to ONE pass: the first net it evaluates)
            f32x4* o = reinterpret_cast<f32x4*>(obs_n_out + d * IN);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (12 * h + 4 * q < IN) o[3 * h + q] = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
        }
    }
    // ---- layer 1 (K = 20 padded to 32: two k-steps)
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
        mfma6x4(w1img + s * 3 * 64, 2 * 3 * 64, lane, x1, x2, x3, a);
        FPV_SPLIT_FENCE();
    }
#pragma unroll
    for (int m = 0; m < (FPV_SPLIT_HAND ? 2 : 4); ++m)       // (hand-placed layer 2: tiles 2 and 3 get theirs between its products)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[m][r] = split_carrier(a[m][r]);
}

#if FPV_SPLIT_HAND
__device__ __forceinline__ void split_net_layer2(const unsigned char* img, int lane, f32x16 (&a)[4], SplitHidden& hd) {
    const int h = lane >> 5;
    const bf16x8* w2img = reinterpret_cast<const bf16x8*>(img);
    const float* bias = reinterpret_cast<const float*>(img + kW2Bytes + kW1Bytes);
    constexpr int kTile = 4 * 2 * 3 * 64, kStep = 3 * 64;       // fragment strides: output tile, k-step (parts are 64 apart)
    f32x16 c[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = bias[128 + (m * 2 + h) * 16 + r];
    bf16x8 w1[4], w2[4], w3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        w3[m] = w2img[m * kTile + 128 + lane]; w1[m] = w2img[m * kTile + lane]; w2[m] = w2img[m * kTile + 64 + lane];
    }
    u32x4 x1, x2, x3;
    {
        float r0, r1;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            uint32_t p1, p2, p3;
            split_cut_a(a[0][2 * pr], a[0][2 * pr + 1], p1, r0, r1);
            split_cut_b(r0, r1, p2, p3);
            x1[pr] = p1; x2[pr] = p2; x3[pr] = p3;
        }
    }
    FPV_PIN();
#define FPV_MF(W, X, m) c[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[m], __builtin_bit_cast(bf16x8, X), c[m], 0, 0, 0)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const bool more = q < 7;
        const int nt = (q + 1) >> 1, ns = (q + 1) & 1;          // next k-step's activations: a[nt][8 ns ..]
        const bf16x8* nw = w2img + (q + 1) * kStep + lane;      // next k-step's fragments (tile m at + m kTile, part at + 64 part)
        u32x4 n1 = x1, n2 = x2, n3 = x3;
        float r0 = 0.f, r1 = 0.f;
        // groups 0 .. 3: products (w3 x1) (w1 x3) (w2 x2) (w2 x1); behind every second product half of pair g's cut
#define FPV_GROUP_CUT(g, W, X)                                                                                                       \
        FPV_MF(W, X, 0); FPV_MF(W, X, 1); FPV_PIN();                                                                                 \
        if (more) { uint32_t p1_; split_cut_a(a[nt][8 * ns + 2 * g], a[nt][8 * ns + 2 * g + 1], p1_, r0, r1); n1[g] = p1_;          \
                    asm volatile("" : "+v"(r0), "+v"(r1)); /* the residuals exist HERE, not where half b uses them */ }              \
        FPV_PIN();                                                                                                                   \
        FPV_MF(W, X, 2); FPV_MF(W, X, 3); FPV_PIN();                                                                                 \
        if (more) { uint32_t p2_, p3_; split_cut_b(r0, r1, p2_, p3_); n2[g] = p2_; n3[g] = p3_; }                                    \
        FPV_PIN();
        FPV_GROUP_CUT(0, w3, x1)
        if (more) {
#pragma unroll
            for (int m = 0; m < 4; ++m) w3[m] = nw[m * kTile + 128];
        }
        FPV_PIN();
        FPV_GROUP_CUT(1, w1, x3)
        FPV_GROUP_CUT(2, w2, x2)
        FPV_GROUP_CUT(3, w2, x1)
        if (more) {
#pragma unroll
            for (int m = 0; m < 4; ++m) w2[m] = nw[m * kTile + 64];
        }
        FPV_PIN();
#undef FPV_GROUP_CUT
        // groups 4, 5: products (w1 x2) (w1 x1); behind each one tanh of layer 1's tiles 2 (k-steps 0, 1) and 3 (k-steps 2, 3)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (g == 0) FPV_MF(w1, x2, m); else FPV_MF(w1, x1, m);
                FPV_PIN();
                if (q < 4) {
                    const int e = (q & 1) * 8 + g * 4 + m;
                    a[2 + (q >> 1)][e] = split_carrier(a[2 + (q >> 1)][e]);
                    FPV_PIN();
                }
            }
        if (more) {
#pragma unroll
            for (int m = 0; m < 4; ++m) w1[m] = nw[m * kTile];
        }
        FPV_PIN();
        x1 = n1; x2 = n2; x3 = n3;
    }
#undef FPV_MF
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = split_carrier(c[m][r]);
#pragma unroll
    for (int m = 0; m < 4; ++m) hd.c[m] = c[m];
}
#else
__device__ __forceinline__ void split_net_layer2(const unsigned char* img, int lane, const f32x16 (&a)[4], SplitHidden& hd) {
    const int h = lane >> 5;
    const bf16x8* w2img = reinterpret_cast<const bf16x8*>(img);
    const float* bias = reinterpret_cast<const float*>(img + kW2Bytes + kW1Bytes);
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
            mfma6x4(w2img + (kt * 2 + s) * 3 * 64, 4 * 2 * 3 * 64, lane, x1, x2, x3, c);
            FPV_SPLIT_FENCE();                   // keep the next k-step's weight fragments from being hoisted here
        }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = split_carrier(c[m][r]);

#pragma unroll
    for (int m = 0; m < 4; ++m) hd.c[m] = c[m];
}
#endif   // FPV_SPLIT_HAND

template <int IN = kIn>
__device__ __forceinline__ void split_net_hidden(const unsigned char* img, const float* nrm, float clip, const float* __restrict__ obs,
                                                 int64_t d, bool live, int lane, float* obs_n_out, SplitHidden& hd) {
    f32x16 a[4];
    split_net_layer1<IN>(img, nrm, clip, obs, d, live, lane, obs_n_out, a);
    split_net_layer2(img, lane, a, hd);
}

// ACT: action words (4 hover, 1 yaw; the head block holds zeros beyond)
// Pre: called in the action net's pass behind its arithmetic and in front of its stores (a caller with LDS-DMA in flight waits there:
// vmcnt(0) before the row stores are issued waits for the DMA alone)
struct SplitNoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <int ACT = 4, class Pre = SplitNoHook>
__device__ __forceinline__ f32x4 split_net_head(int net, const float* head, const SplitHidden& hd, int64_t d, bool live, int lane,
                                                const SplitOut& out, const float* eps_in, uint64_t tick, bool deterministic,
                                                uint64_t seed, uint64_t id_offset, Pre pre = Pre()) {
    const int h = lane >> 5;
    const f32x16 (&c)[4] = hd.c;
    f32x4 cl = f32x4{0.f, 0.f, 0.f, 0.f};
    // The two lane halves are combined with half_sum() (v_permlane32_swap, VALU only).  The first version went
    // through __shfl_xor -> ds_bpermute_b32 and about 1 row in 3 000 came out with a wrong FIRST head component
    // (1 311 bad rows in 3.9 x 10^6 with -DFPV_HALFSUM_BPERMUTE, 0 in 2 x 10^7 with the VALU swap; tools/stress_policy.py;
    // cause unproven).
    if (net == 0) {
        // ---- action head on the VALU, sampling epilogue
        // (the four outputs as two packed pairs: a hidden unit costs two v_pk_fma_f32 instead of four v_fmac_f32 — the same fused
        // multiply-adds in the same order, so the same bits —, and a tile's sixteen weight quads are requested before the first is used)
#if FPV_SPLIT_HEAD_PACKED
        f32x2 mu01 = f32x2{0.f, 0.f}, mu23 = f32x2{0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = *reinterpret_cast<const f32x4*>(head + (((h * 4 + kt) * 16 + r) * 4));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x2 v = f32x2{c[kt][r], c[kt][r]};
                mu01 = __builtin_elementwise_fma(f32x2{w[r].x, w[r].y}, v, mu01);
                mu23 = __builtin_elementwise_fma(f32x2{w[r].z, w[r].w}, v, mu23);
            }
        }
        float mu[4] = {mu01.x, mu01.y, mu23.x, mu23.y};
#else
        float mu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(head + (((h * 4 + kt) * 16 + r) * 4));
                const float v = c[kt][r];
                mu[0] += w.x * v; mu[1] += w.y * v; mu[2] += w.z * v; mu[3] += w.w * v;
            }
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) mu[j] = half_sum(mu[j]) + head[640 + j];
        float eps[4] = {0.f, 0.f, 0.f, 0.f};
        if (!deterministic) {
            if (eps_in) {
                if constexpr (ACT == 4) {
                    const f32x4 e = live ? reinterpret_cast<const f32x4*>(eps_in)[d] : f32x4{0.f, 0.f, 0.f, 0.f};
                    eps[0] = e.x; eps[1] = e.y; eps[2] = e.z; eps[3] = e.w;
                } else {
#pragma unroll
                    for (int j = 0; j < ACT; ++j) eps[j] = live ? eps_in[d * ACT + j] : 0.f;
                }
            } else {
                uint32_t xr[4];
                philox4x32_10((uint32_t)(id_offset + (uint64_t)d), (uint32_t)tick, (uint32_t)(tick >> 32), 0x504F4C31u,
                             (uint32_t)seed, (uint32_t)(seed >> 32), xr);
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
            act[j] = __builtin_fmaf(head[652 + j], eps[j], mu[j]);      // (explicit fmas: the same instructions in every kernel this
            if (j < ACT) lp = __builtin_fmaf(-0.5f * eps[j], eps[j], lp);   // function is inlined into, whatever contraction would choose)
        }
        cl = f32x4{fminf(fmaxf(act[0], -1.f), 1.f), fminf(fmaxf(act[1], -1.f), 1.f), fminf(fmaxf(act[2], -1.f), 1.f),
                   fminf(fmaxf(act[3], -1.f), 1.f)};
        pre();
        if (live) {
            if (h == 0) {
                if constexpr (ACT == 4) {
                    if (out.action) reinterpret_cast<f32x4*>(out.action)[d] = f32x4{act[0], act[1], act[2], act[3]};
                    if (out.clipped) reinterpret_cast<f32x4*>(out.clipped)[d] = cl;
                } else {
#pragma unroll
                    for (int j = 0; j < ACT; ++j) {
                        if (out.action) out.action[d * ACT + j] = act[j];
                        if (out.clipped) out.clipped[d * ACT + j] = cl[j];
                    }
                }
            } else {
                if (out.logp) out.logp[d] = lp;
                if constexpr (ACT == 4) {
                    if (out.eps_out) reinterpret_cast<f32x4*>(out.eps_out)[d] = f32x4{eps[0], eps[1], eps[2], eps[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < ACT; ++j)
                        if (out.eps_out) out.eps_out[d * ACT + j] = eps[j];
                }
            }
        }
    } else {
#if FPV_SPLIT_HEAD_PACKED
        // (two partial sums, even and odd hidden units of a quad pair, as packed multiply-adds: 32 instead of 64 instructions)
        f32x2 vv = f32x2{0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = *reinterpret_cast<const f32x4*>(head + 512 + (h * 4 + kt) * 16 + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vv = __builtin_elementwise_fma(f32x2{w[q].x, w[q].y}, f32x2{c[kt][4 * q], c[kt][4 * q + 1]}, vv);
                vv = __builtin_elementwise_fma(f32x2{w[q].z, w[q].w}, f32x2{c[kt][4 * q + 2], c[kt][4 * q + 3]}, vv);
            }
        }
        float v = vv.x + vv.y;
#else
        float v = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) v += head[512 + (h * 4 + kt) * 16 + r] * c[kt][r];
#endif
        v = half_sum(v);
        if (live && h == 1) out.value[d] = v + head[644];
    }
    return cl;
}

template <int IN = kIn, int ACT = 4>
__device__ __forceinline__ f32x4 split_net_row(int net, const unsigned char* img, const float* head, const float* nrm, float clip,
                                               const float* __restrict__ obs, int64_t d, bool live, int lane, const SplitOut& out,
                                               const float* eps_in, uint64_t tick, bool deterministic, uint64_t seed, uint64_t id_offset) {
    SplitHidden hd;
    split_net_hidden<IN>(img, nrm, clip, obs, d, live, lane, out.obs_n, hd);
    return split_net_head<ACT>(net, head, hd, d, live, lane, out, eps_in, tick, deterministic, seed, id_offset);
}

// this net's split image -> LDS (all requests in flight before the first LDS write); THREADS = workgroup size.  In two halves for a
// caller that has other work to do while the requests are under way: split_stage_request (global -> registers) and
// split_stage_commit (registers -> LDS; the image's space must be free by then).
template <int THREADS, int OFF = 0, int BYTES = kNetBytes>      // bytes [OFF, OFF + BYTES) of a net's image
struct SplitStage {
    static constexpr int kGran = BYTES / 16, kPer = (kGran + THREADS - 1) / THREADS;
    static_assert(OFF % 16 == 0 && BYTES % 16 == 0, "16-byte granules");
    f32x4 v[kPer];
};
template <int THREADS, int OFF, int BYTES>
__device__ __forceinline__ void split_stage_request(const unsigned char* __restrict__ image, int net, SplitStage<THREADS, OFF, BYTES>& st) {
    using S = SplitStage<THREADS, OFF, BYTES>;
    const f32x4* src = reinterpret_cast<const f32x4*>(image + (size_t)net * kNetBytes + OFF);
#pragma unroll
    for (int q = 0; q < S::kPer; ++q) {
        const int g = q * THREADS + threadIdx.x;
        st.v[q] = src[g < S::kGran ? g : 0];
    }
}
template <int THREADS, int OFF, int BYTES>
__device__ __forceinline__ void split_stage_commit(const SplitStage<THREADS, OFF, BYTES>& st, unsigned char* lds_raw) {
    using S = SplitStage<THREADS, OFF, BYTES>;
    f32x4* dst = reinterpret_cast<f32x4*>(lds_raw + OFF);
#pragma unroll
    for (int q = 0; q < S::kPer; ++q) {
        const int g = q * THREADS + threadIdx.x;
        if (g < S::kGran) dst[g] = st.v[q];
    }
}
template <int THREADS>
__device__ __forceinline__ void split_stage_image(const unsigned char* __restrict__ image, int net, unsigned char* lds_raw) {
    SplitStage<THREADS> st;
    split_stage_request(image, net, st);
    split_stage_commit(st, lds_raw);
}

// The same image by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers, no ds_write pass), for a caller that
// has other work for the flight time.  Inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write and drains
// vmcnt(0) in front of the next ds_read (cdna_hip_programming.md §5); an asm statement is outside its bookkeeping, so the caller
// waits itself — s_waitcnt vmcnt(0), then a workgroup barrier — before anyone reads the image.  lds_base: the image's LDS byte address.
template <int THREADS>
__device__ __forceinline__ void split_stage_dma(const unsigned char* __restrict__ image, int net, unsigned lds_base) {
    static_assert(kNetBytes % 1024 == 0, "whole 1-KiB pieces");
    constexpr int kPieces = kNetBytes / 1024, kWaves = THREADS / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* src = image + (size_t)net * kNetBytes + lane * 16;
#pragma unroll
    for (int i = 0; i < (kPieces + kWaves - 1) / kWaves; ++i) {
        const int piece = i * kWaves + wave;                       // wave-uniform
        if (piece < kPieces) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)piece * 1024u);
            const unsigned char* s = src + piece * 1024;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
        }
    }
}

}  // namespace fpv
