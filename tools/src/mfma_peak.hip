// Sustained rate of back-to-back v_mfma_f32_32x32x2_f32 on every SIMD (the ceiling the policy kernels are priced against):
// N waves per SIMD, four independent accumulators per wave, no memory traffic.  Prints TFLOP/s for 1 and 2 waves per SIMD
// and for a duration comparable to the policy kernel (~50 us) as well as a long run (clock settles under sustained load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// the same four-accumulator loop on operands with full-entropy mantissas that change every product (a realistic
// toggle rate: power, hence the clock the chip sustains, depends on the data)
__global__ __launch_bounds__(512) void mfma_loop_random(float* out, int iters, unsigned seed) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) {
        s = s * 1664525u + 1013904223u; a[j] = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;   // [-0.25, 0.25)
        s = s * 1664525u + 1013904223u; b[j] = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;
    }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c3, 0, 0, 0);
        const float t = a[0]; a[0] = b[1]; b[1] = a[2]; a[2] = b[3]; b[3] = a[1]; a[1] = b[2]; b[2] = a[3]; a[3] = b[0]; b[0] = t;
    }
    float r = 0.f;
    for (int k = 0; k < 16; ++k) r += c0[k] + c1[k] + c2[k] + c3[k];
    if (r == 123.456f) out[threadIdx.x] = r;
}

__global__ __launch_bounds__(512) void mfma_chain(float* out, int iters, float a, float b) {
    f32x16 c0 = {};
    for (int i = 0; i < iters; ++i) {       // ONE accumulator: every product depends on the previous one
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        for (int iters : {200, 2000, 20000}) {
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
            hipDeviceSynchronize();
            const int launches = iters >= 20000 ? 5 : 50;
            hipEventRecord(e0);
            for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / launches;
            const double flop = (double)cus * (threads / 64) * iters * 4.0 * 4096.0;
            printf("{\"cus\": %d, \"waves_per_simd\": %d, \"mfma_per_wave\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"implied_GHz\": %.2f}\n",
                   cus, threads / 256, iters * 4, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 2.4);
        }
    }
    for (int iters : {200, 2000, 20000}) {
        const int threads = 512, launches = iters >= 20000 ? 5 : 50;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop_random, dim3(cus), dim3(threads), 0, 0, out, iters, 7u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_loop_random, dim3(cus), dim3(threads), 0, 0, out, iters, 7u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / launches, flop = (double)cus * (threads / 64) * iters * 4.0 * 4096.0;
        printf("{\"random_operands\": true, \"waves_per_simd\": 2, \"mfma_per_wave\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"implied_GHz\": %.2f}\n",
               iters * 4, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 2.4);
    }
    for (int threads : {256, 512}) {
        const int iters = 2000, launches = 50;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_chain, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_chain, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / launches, flop = (double)cus * (threads / 64) * iters * 4.0 * 4096.0;
        printf("{\"dependent_chain\": true, \"waves_per_simd\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f}\n", threads / 256, us, flop / us / 1e6);
    }
    return 0;
}
