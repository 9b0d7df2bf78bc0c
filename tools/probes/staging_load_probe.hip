// round 5 micro-benchmark: cost of the Winograd kernels' input-staging loads on the CU's vector-memory path.
//   hipcc --offload-arch=gfx950 -O3 -o staging_load_probe.bin staging_load_probe.hip
// 8 waves per CU (512 threads, one workgroup per CU, 256 workgroups) each issue `NL` loads per iteration of one of these patterns over a
// [C][176][174] float tensor (a 172-px padded plane), then wait for them:
//   0 dense     : 64 lanes x 8 B contiguous (the A-operand loads)
//   1 rows9x8   : 18 x 18 image rows, 9 lanes x 8 B per row, rows 696 B apart (today's staging: ~7 rows per instruction)
//   2 rows5x16  : the same image as 4 x 16 B + 1 x 8 B per row (16-byte loads, 8-byte aligned)
//   3 rows9x8 with both sub-regions of a workgroup side by side (34-float rows, 17 lanes x 8 B)
// Reports cycles per load instruction per CU (all 8 waves issuing) -- what a staging instruction costs the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PAT>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Wp = 174, plane = 176 * 174;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // every iteration a different region / channel so that nothing is L1-resident (L2 / HBM like the real thing)
        const int region = (blockIdx.x * 37 + it * 11 + wave) % 100;
        const int y0 = (region / 10) * 16, x0 = (region % 10) * 16;
        const float* base = src + (size_t)((blockIdx.x + it * 7 + wave * 3) % 48) * plane + (size_t)y0 * Wp + x0;
        float4 v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (PAT == 0) {
                const float2 t = *reinterpret_cast<const float2*>(base + (j * 64 + lane) * 2);
                v[j] = make_float4(t.x, t.y, 0.f, 0.f);
            } else if (PAT == 1) {
                int e = 64 * (j % 3) + lane; e = e < 162 ? e : 161;
                const int row = e / 9, col = 2 * (e % 9);
                const float2 t = *reinterpret_cast<const float2*>(base + (size_t)(j / 3) * plane + row * Wp + col);
                v[j] = make_float4(t.x, t.y, 0.f, 0.f);
            } else if (PAT == 2) {
                int e = 64 * (j % 2) + lane; e = e < 90 ? e : 89;                   // 2 instructions per channel (90 lane tasks), 3 channels
                const int row = e / 5, k5 = e % 5;
                const float* p = base + (size_t)(j / 2) * plane + row * Wp + 4 * k5;
                if (k5 < 4) { typedef float f4u __attribute__((ext_vector_type(4), aligned(8))); const f4u t = *reinterpret_cast<const f4u*>(p); v[j] = make_float4(t.x, t.y, t.z, t.w); }
                else { const float2 t = *reinterpret_cast<const float2*>(p); v[j] = make_float4(t.x, t.y, 0.f, 0.f); }
            } else {
                int e = 64 * j + lane; e = e < 306 ? e : 305;                        // one channel: 18 rows x 17 pairs = 306 (4.8 instructions)
                const int row = e / 17, col = 2 * (e % 17);
                const float2 t = *reinterpret_cast<const float2*>(base + row * Wp + col);
                v[j] = make_float4(t.x, t.y, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    const size_t n = (size_t)48 * 176 * 174 + 4096;
    (void)hipMalloc(&src, n * 4); (void)hipMemset(src, 0, n * 4);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 300;
    static unsigned long long h[256 * 8];
    auto report = [&](const char* name, double bytes_per_instr) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256 * 8; ++i) m += h[i]; m /= 256.0 * 8;
        const double per_cu = m / (iters * 6.0 * 8.0);          // 8 waves x 6 loads per iteration share the CU's path
        printf("%-28s %.1f cycles per load instruction per CU, %.1f useful B / clk / CU\n", name, per_cu, bytes_per_instr / per_cu);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, src, out, cyc, iters); report("dense 64 x 8 B", 512);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, src, out, cyc, iters); report("rows 9 x 8 B (today)", 64 * 8 * 162.0 / 192);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, src, out, cyc, iters); report("rows 4 x 16 B + 8 B", 18 * 72.0 / 2);
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, src, out, cyc, iters); report("rows 17 x 8 B (pair)", 18 * 136.0 / 4.8);
    }
    return 0;
}
