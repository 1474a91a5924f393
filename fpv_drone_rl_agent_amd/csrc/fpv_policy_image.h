// fpv_policy_image.h — layout of the packed actor-critic weight image (written by policy_pack_kernel, fpv_policy.hip) for
// the v_mfma_f32_32x32x2_f32 kernels that keep it in LDS: fpv_policy.hip and fpv_rollstep.hip.
#pragma once

namespace fpv {

constexpr int kHid = 128, kIn = 20, kInPad = 24, kAct = 4;
constexpr int kNetFloats = 4 * 12 * 64 + 128 + 4 * 4 * 4 * 64 * 4 + 128;  // 19712
constexpr int kOffW1 = 0, kOffB1 = 3072, kOffW2 = 3200, kOffB2 = 19584;
constexpr int kHeadBase = 2 * kNetFloats;                                   // 39424
constexpr int kOffWa = kHeadBase, kOffWv = kHeadBase + 512, kOffBa = kHeadBase + 640, kOffBv = kHeadBase + 644,
              kOffLogStd = kHeadBase + 648, kOffStd = kHeadBase + 652, kOffLogpConst = kHeadBase + 656;
constexpr int kImageFloats = kHeadBase + 660;                               // packed image
// LDS: the staged nets, then the 660 head floats, then the normaliser mean[20] / inv_std[20] (160 496 B for both nets)

}  // namespace fpv
