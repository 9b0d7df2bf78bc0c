// round 5 micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 for ONE wave per SIMD under different accumulator orders
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_16x16_order_probe.bin mfma_f32_16x16_order_probe.hip && ./mfma_f32_16x16_order_probe.bin
// pattern 0: 72 accumulators round robin (dependent distance 72)        1: groups (a0 a1 a0 a1) -- the F(4x4) kernel's order (distance 2)
// pattern 2: back-to-back pairs (a0 a0 a1 a1)                           3: distance 4 (a0 a1 a2 a3 a0 a1 a2 a3)
// pattern 4: as 1 with accumulators in VGPRs only (32 accumulators)      5: 32x32x2 round robin over 16 accumulators (reference)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[72];
    f32x4 av[8];
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < 72; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 72; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if (PAT == 1 || PAT == 4) {
            constexpr int NA = PAT == 4 ? 16 : 36;
#pragma unroll
            for (int rep = 0; rep < 36 / NA; ++rep)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 * i + 1], 0, 0, 0);
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[2 * i + 1], 0, 0, 0);
            }
        } else if (PAT == 6 || PAT == 7) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2 * i + 1], 0, 0, 0);
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[2 * i + 1], 0, 0, 0);
            }
            if (PAT == 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(av[2 * i]) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(av[2 * i + 1]) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(av[2 * i]) : "v"(b), "v"(a));
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(av[2 * i + 1]) : "v"(b), "v"(a));
                }
            }
        } else if (PAT == 2) {
#pragma unroll
            for (int i = 0; i < 72; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            }
        } else if (PAT == 3) {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r ? b : a, r ? a : b, acc[4 * i + j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 72; ++i) s += acc[i];
    asm volatile("s_nop 15\n s_nop 15");
#pragma unroll
    for (int i = 0; i < 8; ++i) s += av[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256, 1) void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[16];
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 200;
    unsigned long long h[256];
    auto report = [&](const char* name, double mf) {
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
        printf("%-58s %.1f cycles per MFMA\n", name, m / (iters * mf));
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 round robin over 72 accumulators", 144);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 groups a0 a1 a0 a1 (distance 2), 72 acc", 144);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 back-to-back pairs a0 a0 a1 a1", 144);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 distance 4 (a0 a1 a2 a3 a0 a1 a2 a3)", 144);
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 groups a0 a1 a0 a1, 32 accumulators", 128);
        hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 64 builtin accumulators (256 regs), groups", 128);
        hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("16x16x4 64 builtin + 8 inline-asm VGPR-form accumulators", 144);
        hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, out, cyc, iters); report("32x32x2 round robin over 16 accumulators", 144);
    }
    return 0;
}
