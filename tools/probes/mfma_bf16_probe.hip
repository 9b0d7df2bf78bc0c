// Probe (GPU box): operand layout and issue rate of v_mfma_f32_32x32x16_bf16 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_layout(const float* A /*[32][16] row m, k*/, const float* B /*[16][32] k, n*/, float* D /*[32][32]*/) {
    const int lane = threadIdx.x, lo = lane & 31, hi = lane >> 5;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)A[lo * 16 + hi * 8 + i]; b[i] = (__bf16)B[(hi * 8 + i) * 32 + lo]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;     // same as 32x32x2 f32
        D[m * 32 + lo] = c[r];
    }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) c[n][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += c[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<float> A(512), B(512), D(1024), R(1024, 0.f);
    for (auto& v : A) v = (float)(rand() % 17 - 8);
    for (auto& v : B) v = (float)(rand() % 13 - 6);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[m * 32 + n] += A[m * 16 + k] * B[k * 32 + n];
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
    printf("layout mismatches: %d of 1024\n", bad);
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<8>, dim3(256 * 2), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 16 * 8.0 * iters * (256 * 2 * 4);
        printf("32x32x16 bf16: %.3f ms -> %.1f TFLOP/s (8 indep acc, 2 waves/SIMD)\n", ms, flop / ms / 1e9);
    }
    // issue rate versus resident waves per SIMD (launch_bounds 256: 1 workgroup = 1 wave on each SIMD of a CU)
    for (int wgs_per_cu : {1, 2, 3, 4}) {
        const int it2 = 20000 * 2 / wgs_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate<8>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, it2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 32 * 32 * 16 * 8.0 * it2 * (256.0 * wgs_per_cu * 4);
            if (rep) printf("32x32x16 bf16, %d wave(s) per SIMD: %.3f ms -> %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
