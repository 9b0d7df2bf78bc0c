// Storage codecs and Sentinel-1 scaling:
//   to_float32 / to_int16   src/tof/tof_downloading.py:64-72, :51-61   (uint16 <-> float32, /65535, trunc)
//   Sentinel-1 preparation  src/download_and_predict_job.py:699-708    (/65535, saturated samples -> the image's
//                           median, convert_to_db job.py:74-89 on both polarisations)
// The raw `.hkl` arrays are uint16: decoding on the device halves the H2D bytes of a tile.
#include "ttc_internal.h"
#include "radix_select.h"

using namespace ttcsel;

namespace {

__global__ void k_u16_to_f32(const unsigned short* __restrict__ in, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i] / 65535.0f;
}
__global__ void k_f32_to_u16(const float* __restrict__ in, long n, unsigned short* __restrict__ out) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned short)truncf(fminf(fmaxf(in[i], 0.f), 1.f) * 65535.0f);
}

// float_to_int16 (job.py:174-180): NaN -> -32768, clip to [-32768, 32767] / precision, * precision, truncate
__global__ void k_f32_to_i16(const float* __restrict__ in, long n, float precision, short* __restrict__ out) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[i];
    if (isnan(x)) x = -32768.0f;
    const float lo = (float)(-32768.0 / (double)precision), hi = (float)(32767.0 / (double)precision);
    x = fminf(fmaxf(x, lo), hi) * precision;
    out[i] = (short)x;
}

// every problem = one image t (both polarisations pooled, as the reference's boolean-mask indexing does)
struct SrcS1 {
    const float* s1; int per_image;
    __device__ int count() const { return per_image; }
    __device__ bool get(int q, int p, float& v) const { v = s1[(long)(q >> 1) * per_image + p]; return true; }
};

__global__ void k_s1_finish(float* __restrict__ s1, const SelState* __restrict__ st, int per_image) {
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= per_image) return;
    const float med = (fkey_inv(st[2 * t].prefix) + fkey_inv(st[2 * t + 1].prefix)) * 0.5f;
    float x = s1[(long)t * per_image + p];
    if (x == 1.0f) x = med;                                  // job.py:703
    x = 10.0f * log10f(x + (float)(1.0 / 65535.0));          // convert_to_db, job.py:86-89 (min_db = 22)
    if (x < -22.0f) x = -22.0f;
    x = (x + 22.0f) / 22.0f;
    s1[(long)t * per_image + p] = fminf(fmaxf(x, 0.f), 1.f);
}

}  // namespace

ttc_status codec_u16_to_f32(ttc_ctx* c, const uint16_t* d_in, int64_t n, float* d_out, hipStream_t s) {
    if (n == 0) return TTC_OK;
    if (!d_in || !d_out || n < 0) return c->fail(TTC_ERR_ARG, "u16_to_float: bad argument");
    hipLaunchKernelGGL(k_u16_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (long)n, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status codec_f32_to_u16(ttc_ctx* c, const float* d_in, int64_t n, uint16_t* d_out, hipStream_t s) {
    if (n == 0) return TTC_OK;
    if (!d_in || !d_out || n < 0) return c->fail(TTC_ERR_ARG, "float_to_u16: bad argument");
    hipLaunchKernelGGL(k_f32_to_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (long)n, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status codec_f32_to_i16(ttc_ctx* c, const float* d_in, int64_t n, float precision, int16_t* d_out, hipStream_t s) {
    if (n == 0) return TTC_OK;
    if (!d_in || !d_out || n < 0 || !(precision > 0.f)) return c->fail(TTC_ERR_ARG, "float_to_int16: bad argument");
    hipLaunchKernelGGL(k_f32_to_i16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, (long)n, precision, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status codec_s1_to_db(ttc_ctx* c, const uint16_t* d_u16, int T, int X, int Y, float* d_out, hipStream_t s) {
    if (!d_u16 || !d_out || T < 1 || T > 64) return c->fail(TTC_ERR_ARG, "s1_to_db: bad argument (T in [1, 64])");
    const int per = X * Y * 2;
    char* ctl = static_cast<char*>(c->scratch_buf("s1_ctl", 4096 + 128 * 256 * 4));
    if (!ctl) return c->fail(TTC_ERR_NOMEM, "s1 scratch");
    SelState* st = reinterpret_cast<SelState*>(ctl);
    unsigned* hist = reinterpret_cast<unsigned*>(ctl + 4096);
    int* nptr = reinterpret_cast<int*>(ctl + 3072);
    TTC_CHECK(codec_u16_to_f32(c, d_u16, (int64_t)T * per, d_out, s));
    TTC_HIP(c, hipMemsetAsync(hist, 0, 128 * 256 * 4, s));
    TTC_HIP(c, hipMemsetAsync(nptr, 0, sizeof(int), s));      // n = per - 0 (ranks are formed on the device)
    hipLaunchKernelGGL(k_sel_init, dim3((2 * T + 63) / 64), dim3(64), 0, s, st, 2 * T, nptr, per, 1, 0, PctList{});
    TTC_HIP(c, radix_select(SrcS1{d_out, per}, st, hist, 2 * T, s));
    hipLaunchKernelGGL(k_s1_finish, dim3((per + 255) / 256, T), dim3(256), 0, s, d_out, st, per);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
