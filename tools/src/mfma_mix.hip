// What the matrix pipe sustains for the policy kernel's instruction mix (v_mfma_f32_16x16x4_f32 with LDS fragment reads
// and vector work between the products), by waves per SIMD.  Variants: 0 bare products, 1 + one ds_read_b128 pair per
// 8 products (prefetched one step ahead), 2 + tanh-like vector work (12 VALU incl. exp + rcp) per 8 products.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MIX>
__global__ __launch_bounds__(1024) void mix_kernel(float* out, unsigned long long* stamps, int iters) {
    __shared__ f32x4 lds[4096];   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = f32x4{0.01f * (i & 7), 0.02f, -0.01f, 0.03f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 c[8];
    for (int t = 0; t < 8; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float b[8];
    for (int u = 0; u < 8; ++u) b[u] = 0.001f * (lane + u);
    f32x4 w0 = lds[lane], w1 = lds[64 + lane];
    float h = 0.f, v = 0.3f + 0.001f * lane;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            f32x4 n0 = w0, n1 = w1;
            if (MIX >= 1) {
                n0 = lds[((it * 8 + st + 1) & 31) * 128 + lane];
                n1 = lds[((it * 8 + st + 1) & 31) * 128 + 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c[2 * (st & 3)] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], b[st], c[2 * (st & 3)], 0, 0, 0);
                c[2 * (st & 3) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], b[st], c[2 * (st & 3) + 1], 0, 0, 0);
            }
            if (MIX >= 2) {
                const float e = __builtin_amdgcn_exp2f(v * 2.885f);
                const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
                h += n0.x * th; h += n0.y * th; h += n0.z * th; h += n0.w * th; h += n1.x * th;
                v = v * 0.999f + 0.0001f;
                asm volatile("" : "+v"(h), "+v"(v) :: "memory");
            }
            w0 = n0; w1 = n1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = h;
    for (int t = 0; t < 8; ++t) s += c[t].x + c[t].y + c[t].z + c[t].w;
    if (s == 123.456f) out[threadIdx.x] = s;
    if (lane == 0) {
        unsigned long long* q = stamps + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 2;
        q[0] = t1 - t0; q[1] = r1 - r0;
    }
}

// The same question for the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, the instruction of fpv_policy_split.hip's 6-term split):
// 32 cycles per product at 32 768 flop.  MIX 0: bare products; 2: the same 12 vector instructions (exp2 + rcp among them) per
// 8 products as above; 3: 24 per 8 products.  If vector instructions co-issue beside bf16 MFMA, MIX 2 / 3 keep MIX 0's rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MIX>
__global__ __launch_bounds__(1024) void mix_bf16_kernel(float* out, unsigned long long* stamps, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 c[2];
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 16; ++j) c[t][j] = 0.f;
    bf16x8 a[2], b[2];
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 8; ++j) { a[t][j] = (__bf16)(0.01f * (lane + j + t)); b[t][j] = (__bf16)(0.02f * (lane - j + t)); }
    float h = 0.f, v = 0.3f + 0.001f * lane, g = 0.1f;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[r & 1], c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[(r + 1) & 1], c[1], 0, 0, 0);
            }
            if (MIX >= 2) {
#pragma unroll
                for (int q = 0; q < (MIX == 3 ? 2 : 1); ++q) {
                    const float e = __builtin_amdgcn_exp2f(v * 2.885f);
                    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
                    h += g * th; h += v * th; h += e * th; h += 0.5f * th; h += 0.25f * th;
                    v = v * 0.999f + 0.0001f;
                    asm volatile("" : "+v"(h), "+v"(v) :: "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
    float s = h;
    for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 16; ++j) s += c[t][j];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (lane == 0) {
        unsigned long long* q = stamps + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 2;
        q[0] = t1 - t0; q[1] = r1 - r0;
    }
}

template <int MIX>
void run_bf16(int cus, float* out, unsigned long long* stamps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        const int iters = 300, launches = 100;   // 300 x 64 products per wave
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(mix_bf16_kernel<MIX>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mix_bf16_kernel<MIX>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[4096 * 2];
        const int wpb = threads / 64, waves = cus * wpb;
        hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        double cyc_max = 0, clk = 0;
        for (int bk = 0; bk < cus; ++bk)
            for (int w = 0; w < wpb; ++w) {
                const double cy = (double)h[(bk * 16 + w) * 2], re = (double)h[(bk * 16 + w) * 2 + 1];
                cyc_max = cy > cyc_max ? cy : cyc_max; clk += cy / re * 0.1 / waves;
            }
        const double us = ms * 1e3 / launches, flop = (double)waves * iters * 64.0 * 32768.0;
        printf("{\"pipe\": \"bf16 32x32x16\", \"mix\": %d, \"valu_per_8_products\": %d, \"waves_per_simd\": %d, \"us_per_launch\": %.1f, "
               "\"tflops_wall\": %.1f, \"pipe_cycles_per_mfma_slowest_wave\": %.2f, \"clock_GHz\": %.3f}\n", MIX, MIX == 0 ? 0 : (MIX == 3 ? 24 : 12),
               threads / 256, us, flop / us / 1e6, cyc_max / (iters * 64.0) / (threads / 256), clk);
    }
}

template <int MIX>
void run(int cus, float* out, unsigned long long* stamps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 1024}) {
        const int iters = 300, launches = 100;   // 300 x 64 = 19 200 products per wave
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(mix_kernel<MIX>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < launches; ++rep) hipLaunchKernelGGL(mix_kernel<MIX>, dim3(cus), dim3(threads), 0, 0, out, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[4096 * 2];
        const int wpb = threads / 64, waves = cus * wpb;
        hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        double cyc_max = 0, cyc_min = 1e30, clk = 0;
        for (int bk = 0; bk < cus; ++bk)
            for (int w = 0; w < wpb; ++w) {
                const double cy = (double)h[(bk * 16 + w) * 2], re = (double)h[(bk * 16 + w) * 2 + 1];
                cyc_max = cy > cyc_max ? cy : cyc_max; cyc_min = cy < cyc_min ? cy : cyc_min; clk += cy / re * 0.1 / waves;
            }
        const double us = ms * 1e3 / launches, flop = (double)waves * iters * 64.0 * 2048.0;
        printf("{\"mix\": %d, \"waves_per_simd\": %d, \"us_per_launch\": %.1f, \"tflops_wall\": %.1f, \"pipe_cycles_per_mfma_slowest_wave\": %.2f, "
               "\"fastest_wave\": %.2f, \"clock_GHz\": %.3f}\n", MIX, threads / 256, us, flop / us / 1e6,
               cyc_max / (iters * 64.0) / (threads / 256), cyc_min / (iters * 64.0) / (threads / 256), clk);
    }
}

int main() {
    float* out; unsigned long long* stamps;
    hipMalloc(&out, 4096 * 4); hipMalloc(&stamps, 4096 * 16);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    run<0>(prop.multiProcessorCount, out, stamps);
    run<1>(prop.multiProcessorCount, out, stamps);
    run<2>(prop.multiProcessorCount, out, stamps);
    run_bf16<0>(prop.multiProcessorCount, out, stamps);
    run_bf16<2>(prop.multiProcessorCount, out, stamps);
    run_bf16<3>(prop.multiProcessorCount, out, stamps);
    return 0;
}
