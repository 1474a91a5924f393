// Diagnostic build of policy_forward_kernel with phase stamps: s_memtime (shader cycles) and s_memrealtime (100 MHz)
// per wave at each phase boundary.  Settles two questions at once: where the non-matrix
// time of a launch goes, and which clock the chip holds while the kernel runs (dt_memtime / dt_memrealtime x 100 MHz).
// The product kernel carries the markers as no-ops; only this translation unit defines FPV_STAMP.  The stamps go to a
// buffer of their own that nothing else reads.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I fpv_drone_rl_agent_amd/csrc tools/src/policy_stamps.hip -o tools/policy_stamps
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

constexpr int kStampSlots = 16;
__device__ unsigned long long g_stamps[1024 * 16 * kStampSlots * 2];

#define FPV_STAMP(k)                                                                                         \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime();   \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {                                                  \
            unsigned long long* q_ = g_stamps + (((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kStampSlots + (k)) * 2; \
            q_[0] = t_;                                                                                      \
            q_[1] = r_;                                                                                      \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)

#include "fpv_policy.hip"
#include "fpv_policy_w16.hip"

using namespace fpv;

static const char* kPhaseLayered[kStampSlots] = {"start", "stage issued + LDS written", "merge prologue + barrier", "obs load + normalise",
                                                 "pi L1 mfma", "pi tanh1", "pi L2 mfma", "pi tanh2", "pi heads",
                                                 "V L1 mfma", "V tanh1", "V L2 mfma", "V tanh2", "V head", "sample + stores", ""};
static const char* kPhasePipelined[kStampSlots] = {"start", "obs + LDS-DMA issued", "wait(obs, pi.W1) + barrier", "normalise",
                                                   "pi L1", "wait(pi.W2) + pi L2", "wait(V.W1) + barrier", "V L1 (+ pi's last pair)",
                                                   "wait(V.W2) + V L2", "V's last pair, sample, stores", "", "", "", "", "", ""};

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    const int settle = argc > 2 ? atoi(argv[2]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    const int img_floats = policy_image_floats();
    std::vector<float> h_obs(n * 20), h_w[13];
    const int wsz[13] = {128 * 20, 128, 128 * 128, 128, 4 * 128, 4, 4, 128 * 20, 128, 128 * 128, 128, 128, 1};
    float* d_w[13];
    for (int i = 0; i < 13; ++i) {
        h_w[i].resize(wsz[i]);
        for (auto& v : h_w[i]) v = (i == 6 ? -0.5f : 0.15f) * rnd();
        hipMalloc(&d_w[i], wsz[i] * 4);
        hipMemcpy(d_w[i], h_w[i].data(), wsz[i] * 4, hipMemcpyHostToDevice);
    }
    for (auto& v : h_obs) v = 2.0f * rnd();
    std::vector<double> h_rms(41, 0.0), h_rows(1024 * 44), h_ret(3, 1.0);
    for (int i = 0; i < 20; ++i) h_rms[20 + i] = 1.0;
    h_rms[40] = 1e4;
    for (auto& v : h_rows) v = rnd();
    float *image, *obs, *obs_n, *action, *clipped, *value, *logp;
    double *rms, *rms_out, *rows, *ret, *ret_out, *snap;
    uint64_t* tick;
    hipMalloc(&image, img_floats * 4); hipMalloc(&obs, n * 80); hipMalloc(&obs_n, n * 80); hipMalloc(&action, n * 16);
    hipMalloc(&clipped, n * 16); hipMalloc(&value, n * 4); hipMalloc(&logp, n * 4);
    hipMalloc(&rms, 41 * 8); hipMalloc(&rms_out, 41 * 8); hipMalloc(&rows, 1024 * 44 * 8); hipMalloc(&ret, 24);
    hipMalloc(&ret_out, 24); hipMalloc(&snap, 44 * 8); hipMalloc(&tick, 8);
    float* image16;
    hipMalloc(&image16, policy16_image_floats() * 4);
    launch_policy_pack(d_w, image, 20, 4, 0);
    launch_policy16_pack(d_w, image16, 20, 4, 0);
    hipMemcpy(obs, h_obs.data(), n * 80, hipMemcpyHostToDevice);
    hipMemcpy(rms, h_rms.data(), 41 * 8, hipMemcpyHostToDevice);
    hipMemcpy(rows, h_rows.data(), 1024 * 44 * 8, hipMemcpyHostToDevice);
    hipMemcpy(ret, h_ret.data(), 24, hipMemcpyHostToDevice);
    hipMemset(tick, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nrows = (int)std::min<int64_t>(n / 256, 1024);
    std::vector<float> ref_action, ref_value, ref_logp;
    for (int variant = 0; variant < 3; ++variant) {   // 0: layer-at-a-time, 1: the same + merge prologue, 2: pipelined
        const bool with_merge = variant == 1, pipelined = variant == 2;
        const char* const* names = pipelined ? kPhasePipelined : kPhaseLayered;
        const int last = pipelined ? 9 : 14;
        MergeArgs mg{};
        if (with_merge) mg = MergeArgs{rows, ret, rms_out, ret_out, snap, (double)n, nrows, 20};
        auto go = [&] {
            if (pipelined)
                return launch_policy16_forward(image16, obs, rms, tick, obs_n, action, clipped, value, logp, nullptr, n, 7, 0, 10.f,
                                               1e-8f, cus, 20, 4, 0);
            return launch_policy_forward(image, obs, rms, nullptr, tick, obs_n, action, clipped, value, logp, nullptr, n, 7, 0,
                                         10.f, 1e-8f, 0, cus, 3, 0, 20, 4, mg, nullptr, 0);
        };
        for (int i = 0; i < 20; ++i) go();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < settle; ++i) go();           // back-to-back on random data: the clock settles
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const hipError_t err = hipGetLastError();
        if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
        // outputs: the pipelined kernel must reproduce the layer-at-a-time kernel's bits
        std::vector<float> h_action(n * 4), h_value(n), h_logp(n);
        hipMemcpy(h_action.data(), action, n * 16, hipMemcpyDeviceToHost);
        hipMemcpy(h_value.data(), value, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h_logp.data(), logp, n * 4, hipMemcpyDeviceToHost);
        double mism = -1;     // max |difference| to the layer-at-a-time kernel's outputs (same fp32 arithmetic, other summation order)
        if (variant == 0) { ref_action = h_action; ref_value = h_value; ref_logp = h_logp; }
        else if (variant == 2) {
            mism = 0;
            for (int64_t i = 0; i < n * 4; ++i) mism = std::max(mism, (double)fabsf(h_action[i] - ref_action[i]));
            for (int64_t i = 0; i < n; ++i) mism = std::max(mism, std::max((double)fabsf(h_value[i] - ref_value[i]), (double)fabsf(h_logp[i] - ref_logp[i])));
        }
        const int wpb = pipelined ? 16 : 8;
        const int blocks = (int)std::min<int64_t>((n + 255) / 256, cus), waves = blocks * wpb;
        std::vector<unsigned long long> st((size_t)1024 * 16 * kStampSlots * 2);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        printf("{\"n\": %lld, \"kernel\": \"%s\", \"merge_prologue\": %d, \"us_per_launch_hip_events\": %.2f, \"max_abs_diff_vs_layered\": %.3g, \"phases\": [",
               (long long)n, pipelined ? "policy_forward_w16_kernel (16 waves, 16x16x4)" : "policy_forward_kernel<512> (8 waves, 32x32x2)", (int)with_merge,
               ms * 1e3 / settle, mism);
        double tot_med = 0;
        for (int k = 1; k <= last; ++k) {
            std::vector<double> dd[3];
            for (int w = 0; w < waves; ++w) {
                const unsigned long long* q = st.data() + (size_t)w * kStampSlots * 2;
                const double v = (double)(q[2 * k] - q[2 * (k - 1)]);
                dd[0].push_back(v);
                dd[1 + ((w % wpb) >= wpb / 2)].push_back(v);
            }
            for (auto& d : dd) std::sort(d.begin(), d.end());
            tot_med += dd[0][dd[0].size() / 2];
            printf("%s{\"phase\": \"%s\", \"cycles_median\": %.0f, \"p10\": %.0f, \"p90\": %.0f, \"median_older_half\": %.0f, \"median_younger_half\": %.0f}",
                   k > 1 ? ", " : "", names[k], dd[0][dd[0].size() / 2], dd[0][dd[0].size() / 10], dd[0][dd[0].size() * 9 / 10],
                   dd[1][dd[1].size() / 2], dd[2][dd[2].size() / 2]);
        }
        std::vector<double> tot, clk, end_lo, end_hi;
        unsigned long long r_first = ~0ull, r_last = 0;
        for (int w = 0; w < waves; ++w) {
            const unsigned long long* q = st.data() + (size_t)w * kStampSlots * 2;
            tot.push_back((double)(q[2 * last] - q[0]));
            clk.push_back((double)(q[2 * last] - q[0]) / (double)(q[2 * last + 1] - q[1]) * 0.1);
            ((w % wpb) >= wpb / 2 ? end_hi : end_lo).push_back((double)(q[2 * last] - q[0]));
            r_first = std::min(r_first, q[1]); r_last = std::max(r_last, q[2 * last + 1]);
        }
        std::sort(tot.begin(), tot.end()); std::sort(clk.begin(), clk.end());
        std::sort(end_lo.begin(), end_lo.end()); std::sort(end_hi.begin(), end_hi.end());
        printf("], \"wave_cycles_total_median\": %.0f, \"wave_cycles_older_half\": %.0f, \"wave_cycles_younger_half\": %.0f, \"sum_of_phase_medians\": %.0f, "
               "\"clock_GHz_median\": %.3f, \"clock_GHz_p10\": %.3f, \"clock_GHz_p90\": %.3f, \"first_stamp_to_last_stamp_us\": %.2f}\n",
               tot[tot.size() / 2], end_lo[end_lo.size() / 2], end_hi[end_hi.size() / 2], tot_med, clk[clk.size() / 2], clk[clk.size() / 10],
               clk[clk.size() * 9 / 10], (double)(r_last - r_first) * 0.01);
        fflush(stdout);
    }
    return 0;
}
