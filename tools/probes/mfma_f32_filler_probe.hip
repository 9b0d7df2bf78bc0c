// round 5 micro-benchmark: what hides under v_mfma_f32_16x16x4_f32 (fp32 inputs) -- per filler kind, for ONE and TWO waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_filler_probe.bin mfma_f32_filler_probe.hip
// Result that shaped conv3x3_wino4.hip: the fp32 matrix instruction runs on the vector ALU's own FMA lanes ("the f32 VECTOR rate"), so a
// wave's VALU instructions do not overlap with ITS OWN or its SIMD partner's fp32 MFMAs -- every VALU instruction adds its issue time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
enum { F_NONE, F_FMA, F_PKFMA, F_DSREAD, F_DSWRITE, F_SALU, F_GLOAD };

template <int KIND, int N, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(float* out, unsigned long long* cyc, int iters, const float* gsrc) {
    __shared__ float lds[8192];
    f32x4 acc[16];
    float v[8]; v2f pv[8];
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const v2f pa = {a, b}, pb = {b, a};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = a + i; pv[i] = v2f{a + i, b + i}; }
    for (int i = threadIdx.x; i < 8192; i += THREADS) lds[i] = a;
    __syncthreads();
    float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int sacc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int q = (i * N + j) & 7;
                if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(a), "v"(b));
                if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[q]) : "v"(pa), "v"(pb));
                if (KIND == F_DSREAD) asm volatile("ds_read_b64 %0, %1" : "=v"(pv[q]) : "v"((int)((threadIdx.x & 63) * 8 + 512 * q)));
                if (KIND == F_DSWRITE) asm volatile("ds_write_b64 %0, %1" :: "v"((int)((threadIdx.x & 63) * 8 + 512 * q)), "v"(pa) : "memory");
                if (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
                if (KIND == F_GLOAD) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(pv[q]) : "v"((int)((threadIdx.x & 63) * 8 + 512 * q)), "s"(gsrc));
            }
        }
        if (KIND == F_DSREAD || KIND == F_GLOAD) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    float r = s.x + s.y + s.z + s.w + sacc;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i] + pv[i].x + pv[i].y + t[i];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc; float* gsrc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8); (void)hipMalloc(&gsrc, 1 << 20);
    (void)hipMemset(gsrc, 0, 1 << 20);
    const int iters = 400;
    static unsigned long long h[256 * 8];
    auto report = [&](const char* name, int nw) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) for (int w = 0; w < nw; ++w) m += h[i * 8 + w]; m /= 256.0 * nw;
        // per SIMD: nw / 4 waves each issue 16 MFMAs per iteration
        printf("%-64s %.1f cycles per MFMA of the SIMD\n", name, m / (iters * 16.0 * (nw / 4)));
    };
#define RUN(kind, n, thr) hipLaunchKernelGGL((k<kind, n, thr>), dim3(256), dim3(thr), 0, 0, out, cyc, iters, gsrc); report(#kind " x " #n " per MFMA, " #thr " threads", thr / 64);
    for (int rep = 0; rep < 2; ++rep) {
        RUN(F_NONE, 0, 256) RUN(F_NONE, 0, 512)
        RUN(F_FMA, 2, 256) RUN(F_FMA, 4, 256) RUN(F_FMA, 2, 512) RUN(F_FMA, 4, 512)
        RUN(F_PKFMA, 2, 256) RUN(F_PKFMA, 4, 256) RUN(F_PKFMA, 2, 512) RUN(F_PKFMA, 4, 512)
        RUN(F_DSREAD, 1, 256) RUN(F_DSREAD, 2, 256) RUN(F_DSREAD, 1, 512) RUN(F_DSREAD, 2, 512)
        RUN(F_DSWRITE, 1, 256) RUN(F_DSWRITE, 1, 512)
        RUN(F_SALU, 2, 256) RUN(F_SALU, 4, 256) RUN(F_SALU, 2, 512) RUN(F_SALU, 4, 512)
        RUN(F_GLOAD, 1, 256) RUN(F_GLOAD, 1, 512)
    }
    return 0;
}
