// Probe (GPU box): MFMA issue rate of the fp32 conv inner loop with ds_read_b32 operands (one per MFMA operand, as in
// conv3x3_mfma.hip) versus ds_read_b128 operands (4 k-steps per read, k-contiguous LDS layout).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_loop(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 14000; i += 256) lds[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    f32x16 acc[2][4];
    for (int g = 0; g < 2; ++g) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.f;
    const float* in_tile = lds;              // 8 planes x 1024
    const float* w_tile = lds + 8192;        // 9 x 8 x 64
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            const float* ibase = in_tile + hi * 1024 + wave * 128 + lo;
            const float* wbase = w_tile + hi * 64 + lo;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
                    float av[2], bv[4];
                    for (int g = 0; g < 2; ++g) av[g] = wbase[(t * 8 + 2 * kp) * 64 + g * 32];
                    for (int j = 0; j < 4; ++j) bv[j] = ibase[(2 * kp) * 1024 + t * 37 + j * 32];
                    for (int g = 0; g < 2; ++g) for (int j = 0; j < 4; ++j)
                        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], bv[j], acc[g][j], 0, 0, 0);
                }
        } else {
            const float4* ibase = reinterpret_cast<const float4*>(in_tile) + hi * 1024 + wave * 128 + lo;   // [hi][q] x 4 k
            const float4* wbase = reinterpret_cast<const float4*>(w_tile) + hi * 64 + lo;                   // [t][hi][cout] x 4 k
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float4 av[2], bv[4];
                for (int g = 0; g < 2; ++g) av[g] = wbase[(t * 2) * 64 + g * 32];
                for (int j = 0; j < 4; ++j) bv[j] = ibase[t * 37 + j * 32];
#pragma unroll
                for (int kp = 0; kp < 4; ++kp)
                    for (int g = 0; g < 2; ++g) for (int j = 0; j < 4; ++j) {
                        const float a = kp == 0 ? av[g].x : kp == 1 ? av[g].y : kp == 2 ? av[g].z : av[g].w;
                        const float b = kp == 0 ? bv[j].x : kp == 1 ? bv[j].y : kp == 2 ? bv[j].z : bv[j].w;
                        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g][j], 0, 0, 0);
                    }
            }
        }
    }
    float s = 0;
    for (int g = 0; g < 2; ++g) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[g][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    float* out; (void)hipMalloc(&out, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400;
    const size_t lds = 60000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_loop<0>, dim3(512), dim3(256), lds, 0, out, iters);
            else hipLaunchKernelGGL(k_loop<1>, dim3(512), dim3(256), lds, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 32 * 32 * 2 * 288.0 * iters * 512 * 4;
            printf("mode %d (%s): %.3f ms -> %.1f TFLOP/s = %.1f %% of 157.3\n", mode, mode ? "ds_read_b128, 4 k-steps per read" : "ds_read_b32 per operand",
                   ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
        }
    // many short workgroups (the conv's real shape: 4248 tiles of 7 chunks) versus few long ones, same total MFMA count
    for (int wgs : {512, 1024, 4248, 4608}) {
        const int it2 = 7;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_loop<1>, dim3(wgs), dim3(256), lds, 0, out, it2);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 32 * 32 * 2 * 288.0 * it2 * wgs * 4;
            printf("b128 loop, %d workgroups x %d iterations: %.3f ms -> %.1f %% of 157.3\n", wgs, it2, ms, flop / ms / 1e9 / 157.3 * 100);
        }
    }
    // residency versus the dynamic LDS request: where does the second workgroup per CU stop fitting?
    for (size_t l2 : {(size_t)60000, (size_t)65536, (size_t)70000, (size_t)75776, (size_t)78000, (size_t)80000, (size_t)81920, (size_t)90000}) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_loop<1>, 256, l2);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_loop<1>, dim3(512), dim3(256), l2, 0, out, 100);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 32 * 32 * 2 * 288.0 * 100 * 512 * 4;
            if (rep) printf("lds %zu B (occupancy API %d): %.3f ms -> %.1f %% of 157.3\n", l2, nb, ms, flop / ms / 1e9 / 157.3 * 100);
        }
    }
    return 0;
}
