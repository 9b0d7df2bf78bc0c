// Small raster steps of process_tile (src/download_and_predict_job.py:641-995) that sit between the stages built elsewhere:
//   Sen2Cor mask clean-up (:688-697), DEM 5x5 median (:713), snow map (:799-821), merging the Sen2Cor mask into the
//   detected clouds (:843-848), the per-date interpolated fraction (:868, :891, :911), the final clip (:994).
#include "ttc_internal.h"

namespace {

constexpr int kMaxT = 32;

// 20 m mask -> 10 m (repeat 2 x 2); walking the dates in order, two consecutive flagged dates are both cleared (in place, so
// a cleared date no longer pairs with the next one)
__global__ void k_sen2cor_clean(const float* __restrict__ clm20, int T, int w20, int h20, float* __restrict__ out) {
    const int X = 2 * w20, Y = 2 * h20;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int x = p / Y, y = p % Y;
    float v[kMaxT];
    for (int t = 0; t < T; ++t) v[t] = clm20[((long)t * w20 + x / 2) * h20 + y / 2];
    for (int i = 1; i < T; ++i)
        if (v[i - 1] + v[i] == 2.0f) { v[i - 1] = 0.f; v[i] = 0.f; }
    for (int t = 0; t < T; ++t) out[(long)t * X * Y + p] = v[t];
}
// scipy.ndimage.median_filter(size = 5), mode 'reflect' (half-sample symmetric)
__device__ __forceinline__ int refl(int i, int n) { while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i; return i; }
__global__ void k_median5(const float* __restrict__ in, int X, int Y, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int x = p / Y, y = p % Y;
    float v[25];
    int n = 0;
    for (int dx = -2; dx <= 2; ++dx)
        for (int dy = -2; dy <= 2; ++dy) v[n++] = in[refl(x + dx, X) * Y + refl(y + dy, Y)];
    for (int i = 1; i < 25; ++i) {
        const float a = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > a) { v[j + 1] = v[j]; --j; }
        v[j + 1] = a;
    }
    out[p] = v[12];
}
// snow_filter (job.py:799-817): NDSI ramp with NIR / blue / blue-red guards, as a flag
__global__ void k_snow_flags(const float* __restrict__ s2, int npix, unsigned char* __restrict__ flags, int* __restrict__ per_img) {
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const float* v = s2 + ((long)t * npix + p) * 10;
        float ndsi = (v[1] - v[8]) / (v[1] + v[8]);
        if (ndsi < 0.10f) ndsi = 0.f;
        if (ndsi > 0.42f) ndsi = 0.42f;
        float pr = (ndsi - 0.1f) / 0.32f;
        if (v[3] < 0.10f) pr = 0.f;
        if (v[3] > 0.35f && pr > 0.f) pr = 1.f;
        if (v[0] < 0.10f) pr = 0.f;
        if (v[0] > 0.22f && pr > 0.f) pr = 1.f;
        if ((v[0] / v[2]) < 0.75f) pr = 0.f;
        const bool f = pr > 0.f;
        flags[(long)t * npix + p] = f;
        c += f;
    }
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&per_img[t], c);
}
// snow = 1 - binary_dilation(mean_t(flags) < 0.7, iterations = 2)   (job.py:820-821)
__global__ void k_snow_rare(const unsigned char* __restrict__ flags, int T, int npix, unsigned char* __restrict__ rare) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int c = 0;
    for (int t = 0; t < T; ++t) c += flags[(long)t * npix + p];
    rare[p] = ((double)c / (double)T) < 0.7;
}
__global__ void k_snow_map(const unsigned char* __restrict__ rare, int X, int Y, unsigned char* __restrict__ snow) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int x = p / Y, y = p % Y;
    bool any = false;
    for (int dx = -2; dx <= 2 && !any; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= X) continue;
        const int r = 2 - abs(dx);
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy >= 0 && yy < Y && rare[xx * Y + yy]) { any = true; break; }
        }
    }
    snow[p] = !any;
}
__global__ void k_merge_clm(float* __restrict__ cloudshad, float* __restrict__ clm, const unsigned char* __restrict__ fcps, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m = clm[i];
    if (fcps && fcps[i]) { m = 0.f; clm[i] = 0.f; }
    cloudshad[i] = fmaxf(cloudshad[i], m);
}
// eq = NaN: count a > 0; otherwise count a == eq
__global__ void k_count_positive(const float* __restrict__ a, int npix, float eq, int* __restrict__ counts) {
    const int t = blockIdx.y;
    const bool pos = eq != eq;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const float v = a[(long)t * npix + p];
        c += pos ? v > 0.f : v == eq;
    }
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[t], c);
}
__global__ void k_divide(float* __restrict__ a, long n, float d) {       // IEEE division (dem / 90, job.py:993), not a reciprocal multiply
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] / d;
}
__global__ void k_clip01(float* __restrict__ a, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = fminf(fmaxf(a[i], 0.f), 1.f);       // np.clip keeps NaN; fminf / fmaxf would not -- none reach this point
}

}  // namespace

ttc_status prep_sen2cor_clean(ttc_ctx* c, const float* d_clm20, int T, int w20, int h20, float* d_out, hipStream_t s) {
    if (!d_clm20 || !d_out || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "sen2cor_clean: bad argument (T in [1, 32])");
    const int npix = 4 * w20 * h20;
    hipLaunchKernelGGL(k_sen2cor_clean, dim3((npix + 255) / 256), dim3(256), 0, s, d_clm20, T, w20, h20, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
ttc_status prep_median5(ttc_ctx* c, const float* d_in, int X, int Y, float* d_out, hipStream_t s) {
    if (!d_in || !d_out || d_in == d_out) return c->fail(TTC_ERR_ARG, "median5: bad argument (not in place)");
    hipLaunchKernelGGL(k_median5, dim3((X * Y + 255) / 256), dim3(256), 0, s, d_in, X, Y, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
ttc_status prep_snow(ttc_ctx* c, const float* d_s2, int T, int X, int Y, uint8_t* d_snow, int32_t* h_per_image, hipStream_t s) {
    if (!d_s2 || !d_snow || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "snow_map: bad argument (T in [1, 32])");
    const int npix = X * Y;
    unsigned char* flags = static_cast<unsigned char*>(c->scratch_buf("prep_snow", (size_t)(T + 1) * npix));
    int* cnt = static_cast<int*>(c->scratch_buf("prep_cnt", sizeof(int) * kMaxT));
    if (!flags || !cnt) return c->fail(TTC_ERR_NOMEM, "snow scratch");
    TTC_HIP(c, hipMemsetAsync(cnt, 0, sizeof(int) * kMaxT, s));
    hipLaunchKernelGGL(k_snow_flags, dim3(32, T), dim3(256), 0, s, d_s2, npix, flags, cnt);
    hipLaunchKernelGGL(k_snow_rare, dim3((npix + 255) / 256), dim3(256), 0, s, flags, T, npix, flags + (size_t)T * npix);
    hipLaunchKernelGGL(k_snow_map, dim3((npix + 255) / 256), dim3(256), 0, s, flags + (size_t)T * npix, X, Y, d_snow);
    TTC_HIP(c, hipGetLastError());
    if (h_per_image) {
        TTC_HIP(c, hipMemcpyAsync(h_per_image, cnt, sizeof(int) * T, hipMemcpyDeviceToHost, s));
        TTC_HIP(c, hipStreamSynchronize(s));
    }
    return TTC_OK;
}
ttc_status prep_merge_clm(ttc_ctx* c, float* d_cloudshad, float* d_clm, const uint8_t* d_fcps, int64_t n, hipStream_t s) {
    if (!d_cloudshad || !d_clm || n < 0) return c->fail(TTC_ERR_ARG, "merge_cloud_masks: bad argument");
    if (n) hipLaunchKernelGGL(k_merge_clm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_cloudshad, d_clm, d_fcps, (long)n);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
ttc_status prep_count_positive(ttc_ctx* c, const float* d_a, int T, int npix, float eq, int32_t* h_counts, hipStream_t s) {
    if (!d_a || !h_counts || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "count_positive: bad argument (T in [1, 32])");
    int* cnt = static_cast<int*>(c->scratch_buf("prep_cnt", sizeof(int) * kMaxT));
    if (!cnt) return c->fail(TTC_ERR_NOMEM, "count scratch");
    TTC_HIP(c, hipMemsetAsync(cnt, 0, sizeof(int) * kMaxT, s));
    hipLaunchKernelGGL(k_count_positive, dim3(32, T), dim3(256), 0, s, d_a, npix, eq, cnt);
    TTC_HIP(c, hipMemcpyAsync(h_counts, cnt, sizeof(int) * T, hipMemcpyDeviceToHost, s));
    TTC_HIP(c, hipStreamSynchronize(s));
    return TTC_OK;
}
ttc_status prep_clip01(ttc_ctx* c, float* d_a, int64_t n, hipStream_t s) {
    if (!d_a || n < 0) return c->fail(TTC_ERR_ARG, "clip01: bad argument");
    if (n) hipLaunchKernelGGL(k_clip01, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_a, (long)n);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
ttc_status prep_divide(ttc_ctx* c, float* d_a, int64_t n, float divisor, hipStream_t s) {
    if (!d_a || n < 0) return c->fail(TTC_ERR_ARG, "divide: bad argument");
    if (n) hipLaunchKernelGGL(k_divide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_a, (long)n, divisor);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
