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

// Round 2: the rotating-operand loop above feeds v_mov results straight into MFMA sources every
// iteration, which can insert VALU->MFMA dependency stalls; this one keeps SIXTEEN independent random A and B operands
// resident in registers (no VALU inside the loop) and walks through them, so consecutive products see different
// full-entropy operands.  It also stamps s_memtime / s_memrealtime around the loop: cycles per MFMA and the clock the
// chip holds are read directly instead of being inferred from wall time.
__global__ __launch_bounds__(512) void mfma_loop_resident(float* out, unsigned long long* stamps, int iters, unsigned seed) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
    float a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        s = s * 1664525u + 1013904223u; a[j] = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;   // [-0.25, 0.25)
        s = s * 1664525u + 1013904223u; b[j] = __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j * 5 + 3) & 15], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 1], b[(j * 5 + 8) & 15], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 2], b[(j * 5 + 13) & 15], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 3], b[(j * 5 + 2) & 15], c3, 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    float r = 0.f;
    for (int k = 0; k < 16; ++k) r += c0[k] + c1[k] + c2[k] + c3[k];
    if (r == 123.456f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* q = stamps + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 2;
        q[0] = t1 - t0; q[1] = r1 - r0;
    }
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
    {   // resident random operands, in-kernel clock
        unsigned long long* stamps;
        hipMalloc(&stamps, 1024 * 8 * 16);
        for (int threads : {256, 512})
            for (int iters : {38, 500, 5000}) {   // 38 x 16 = 608 products per wave: the policy kernel's count
                const int launches = iters >= 5000 ? 40 : 400;
                for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(mfma_loop_resident, dim3(cus), dim3(threads), 0, 0, out, stamps, iters, 7u);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mfma_loop_resident, dim3(cus), dim3(threads), 0, 0, out, stamps, iters, 7u + rep);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2048 * 2];
                hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * cus * (threads / 64), hipMemcpyDeviceToHost);
                double cyc = 0, real = 0;
                const int waves = cus * (threads / 64);
                for (int w = 0; w < waves; ++w) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
                const double us = ms * 1e3 / launches, flop = (double)cus * (threads / 64) * iters * 16.0 * 4096.0;
                printf("{\"resident_random_operands\": true, \"waves_per_simd\": %d, \"mfma_per_wave\": %d, \"us_per_launch\": %.1f, "
                       "\"tflops_wall\": %.1f, \"cycles_per_mfma_per_simd\": %.2f, \"in_kernel_clock_GHz\": %.3f, \"tflops_in_loop\": %.1f}\n",
                       threads / 256, iters * 16, us, flop / us / 1e6, cyc / waves / (iters * 16.0) / (threads / 256),
                       cyc / real * 0.1, flop / (real / waves * 0.01) / 1e6);
            }
    }
    return 0;
}
