// fpv_mlp.h — pieces shared by the MFMA kernels of the actor-critic (fpv_policy.hip, fpv_policy_split.hip, fpv_ppo.hip):
// accumulator-tile bookkeeping of v_mfma_f32_32x32x2_f32 in the transposed formulation (H^T = W . X^T), tanh, Philox.
#pragma once
#include "fpv_dynamics.h"   // philox4x32_10
#include "fpv_internal.h"

// Phase stamps (s_memtime / s_memrealtime) exist only in the diagnostic build tools/src/policy_stamps.hip, which defines
// FPV_STAMP before including the kernel source; in the product the markers compile to nothing.
#ifndef FPV_STAMP
#define FPV_STAMP(k) ((void)0)
#endif

namespace fpv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row of a 32 x 32 accumulator tile held by register r of a lane in half h (lane >> 5): rows come in quads, so
// four consecutive registers are four consecutive rows and a register quad is one 16-byte store.
__host__ __device__ constexpr int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// one hidden layer's 64 values per lane: 4 accumulator tiles
struct Hidden {
    f32x16 t[4];
};

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}); v_exp_f32 is 2^x.  |error| < 3e-7 absolute.
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

// tanh of a pre-activation that arrives already multiplied by 2 log2(e): the exact-fp32 policy image (fpv_policy.hip's
// pack kernel) carries that factor in the hidden layers' weights and biases, which takes one multiply out of every tanh —
// vector instructions do not overlap the fp32 MFMA stream, so each one is launch time.
constexpr float kTanhPrescale = 2.8853900817779268f;
// FPV_TANH_FOLD: tanh(x) = 1 - 2 r with r = 1 / (1 + e^{2x}), and the layer that consumes tanh's output is affine in it:
// W (1 - 2 r) + b = (b + sum_k W_k) - 2 W r.  The image then carries -2 W and b + sum_k W_k for the second hidden layer and the
// heads (policy_pack_kernel: the sum in double, the factor 2 exact), the kernels hand r from layer to layer, and every tanh is
// three vector instructions instead of four — 256 fewer per wave, on a pipe that the matrix products share with them.  The chain
// stays a k-ordered fp32 fma chain; its partial sums start at b + sum W instead of b (the same magnitude for any weights whose
// row sums are not far above their row norms), so the results differ from the unfolded form at rounding level only
// (tests/test_policy_gpu.py holds both to the same bounds against an fp64 evaluation).
#ifndef FPV_TANH_FOLD
#define FPV_TANH_FOLD 1
#endif
// What a hidden layer hands to the next one ("the carrier"): with the fold r = (1 - tanh x) / 2, NOT tanh x — only a consumer whose
// image was packed by policy_pack_kernel of the same build (-2 W, b + sum W) may take it; without the fold tanh x itself.
__device__ __forceinline__ float tanh_carrier_prescaled(float z) {
#if FPV_TANH_FOLD
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));          // r; the consumer's image holds -2 W and b + sum W
#else
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
#endif
}

// bias image [m][h][16] -> accumulator tiles
__device__ __forceinline__ void bias_init(Hidden& a, const float* __restrict__ lb, int h) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4* b = reinterpret_cast<const f32x4*>(lb + (m * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[q];
            a.t[m][4 * q] = v.x; a.t[m][4 * q + 1] = v.y; a.t[m][4 * q + 2] = v.z; a.t[m][4 * q + 3] = v.w;
        }
    }
}

__device__ __forceinline__ void carrier_all_prescaled(Hidden& a) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.t[m][r] = tanh_carrier_prescaled(a.t[m][r]);
}

__device__ __forceinline__ void tanh_all(Hidden& a) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) a.t[m][r] = fast_tanh(a.t[m][r]);
}

}  // namespace fpv
