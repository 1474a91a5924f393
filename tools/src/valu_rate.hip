// VALU issue rate on gfx950 by instruction and waves per SIMD: v_fma_f32 against v_pk_fma_f32 (two fp32 lanes per
// register pair) and v_exp_f32.  Decides whether packing two drones per lane can shorten the env-step kernel, which is
// bound by single-wave vector issue at 65 536 drones (profiles/r02_step_instruction_mix.json).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* stamps, int iters) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.001f * (threadIdx.x + i); p[i] = f32x2{a[i], a[i] + 1.0f}; }
    const float m = 1.0001f, c = 0.0003f;
    const f32x2 pm = {1.0001f, 0.9999f}, pc = {0.0003f, -0.0002f};
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // inline asm: hipcc would otherwise pack the scalar chain into v_pk_fma_f32 itself (SLP)
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int cus, float* out, unsigned long long* stamps) {
    for (int threads : {256, 512, 1024}) {
        const int iters = 2000;
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipDeviceSynchronize();
        static unsigned long long h[4096];
        hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < threads / 64; ++w) mx = h[b * 16 + w] > mx ? h[b * 16 + w] : mx;
        printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instruction_per_simd\": %.2f}\n", name, threads / 256,
               mx / (iters * 64.0) / (threads / 256));
    }
}

int main() {
    float* out; unsigned long long* stamps;
    hipMalloc(&out, 4096 * 4); hipMalloc(&stamps, 4096 * 8);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    run<0>("v_fma_f32", prop.multiProcessorCount, out, stamps);
    run<1>("v_pk_fma_f32", prop.multiProcessorCount, out, stamps);
    run<2>("v_exp_f32", prop.multiProcessorCount, out, stamps);
    return 0;
}
