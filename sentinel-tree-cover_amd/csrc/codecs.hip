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

// Sentinel-1 preparation (job.py:699-708): the per-image median (both polarisations pooled, as the reference's boolean-mask
// indexing does) replaces saturated samples, then convert_to_db.  The median is selected on the STORED uint16 values --
// x = u / 65535 is monotone in u, so the order statistics of x are x(order statistics of u) -- with two 8-bit radix passes over
// 2 bytes per sample; both middle order statistics come out of the same two histograms.  (Round 2 decoded to float32 first
// and ran the generic 4-pass float select twice per image: 8 passes over 4-byte samples + a read-modify-write finish.)
struct S1Sel { unsigned prefix[2]; long long k[2]; };     // [lower, upper] middle order statistic of one image
__global__ void k_s1_hist(const uint16_t* __restrict__ u16, int per_image, const S1Sel* __restrict__ sel, int pass,
                          unsigned* __restrict__ hist /*[T][2][256]*/) {
    __shared__ unsigned h[2][256];
    const int t = blockIdx.y;
    h[0][threadIdx.x] = 0; h[1][threadIdx.x] = 0;
    __syncthreads();
    const uint16_t* src = u16 + (long)t * per_image;
    const unsigned p0 = pass ? sel[t].prefix[0] : 0u, p1 = pass ? sel[t].prefix[1] : 0u;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < per_image; p += gridDim.x * blockDim.x) {
        const unsigned u = src[p];
        if (!pass) atomicAdd(&h[0][u >> 8], 1u);                       // pass 0: one histogram serves both ranks
        else {
            if ((u >> 8) == p0) atomicAdd(&h[0][u & 255u], 1u);
            if ((u >> 8) == p1) atomicAdd(&h[1][u & 255u], 1u);
        }
    }
    __syncthreads();
    if (h[0][threadIdx.x]) atomicAdd(&hist[(t * 2 + 0) * 256 + threadIdx.x], h[0][threadIdx.x]);
    if (pass && h[1][threadIdx.x]) atomicAdd(&hist[(t * 2 + 1) * 256 + threadIdx.x], h[1][threadIdx.x]);
}
// one wave per (image, rank): walk the 256 bins to the one holding the rank; clears the histogram for the next pass
__global__ void k_s1_pick(S1Sel* __restrict__ sel, int per_image, int pass, unsigned* __restrict__ hist) {
    const int t = blockIdx.x, which = blockIdx.y, lane = threadIdx.x;
    const unsigned* h = hist + (t * 2 + (pass ? which : 0)) * 256;
    unsigned c[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = h[4 * lane + j]; mine += c[j]; }
    unsigned incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const long long excl = (long long)incl - mine;
    const long long k = pass ? sel[t].k[which] : (which ? (long long)per_image / 2 : ((long long)per_image - 1) / 2);
    const bool here = k >= excl && k < (long long)incl;
    const unsigned long long m = __ballot(here);
    const int owner = m ? __ffsll((long long)m) - 1 : 63;
    if (lane == owner) {
        long long r = k - excl;
        int b = 0;
        for (; b < 3; ++b) { if (r < (long long)c[b]) break; r -= c[b]; }
        const unsigned digit = (unsigned)(4 * lane + b);
        sel[t].prefix[which] = pass ? ((sel[t].prefix[which] << 8) | digit) : digit;
        sel[t].k[which] = m ? r : 0;
    }
}
__global__ void k_s1_clear(unsigned* __restrict__ hist, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hist[i] = 0;
}
// am / X / Y: the result is written on the [X, Y] grid of the 20 m stack; adjust_shape (job.py:260-310) runs AFTER the dB conversion in the
// reference (:699-718), so the median above is the one of the image AS STORED and only this pass re-indexes
__global__ void k_s1_finish(const uint16_t* __restrict__ u16, const S1Sel* __restrict__ sel, int per_image, AdjustMap am, int X, int Y,
                            float* __restrict__ s1) {
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y * 2) return;
    const float med = ((float)sel[t].prefix[0] / 65535.0f + (float)sel[t].prefix[1] / 65535.0f) * 0.5f;   // np.median of the float32 image
    const int pix = p >> 1, i = pix / Y, j = pix - i * Y;
    const int is = min(max(i + am.o1, 0), am.n1 - 1), js = min(max(j + am.o2, 0), am.n2 - 1);
    float x = (float)u16[(long)t * per_image + ((long)is * am.n2 + js) * 2 + (p & 1)] / 65535.0f;   // to_float32, tof_downloading.py:64-72
    if (x == 1.0f) x = med;                                  // job.py:703
    x = 10.0f * log10f(x + (float)(1.0 / 65535.0));          // convert_to_db, job.py:86-89 (min_db = 22)
    if (x < -22.0f) x = -22.0f;
    x = (x + 22.0f) / 22.0f;
    s1[(long)t * X * Y * 2 + p] = fminf(fmaxf(x, 0.f), 1.f);
}

// adjust_shape (job.py:260-310) on a float32 array [T, n1, n2, C] -> [T, X, Y, C]
__global__ void k_adjust_shape(const float* __restrict__ in, AdjustMap am, int X, int Y, int C, float* __restrict__ out) {
    const int t = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)X * Y * C) return;
    const int ch = (int)(p % C);
    const int pix = (int)(p / C), i = pix / Y, j = pix - i * Y;
    const int is = min(max(i + am.o1, 0), am.n1 - 1), js = min(max(j + am.o2, 0), am.n2 - 1);
    out[(long)t * X * Y * C + p] = in[(((long)t * am.n1 + is) * am.n2 + js) * C + ch];
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

ttc_status codec_adjust_shape(ttc_ctx* c, const float* d_in, int T, int n1, int n2, int C, int width, int height, float* d_out, hipStream_t s) {
    AdjustMap am;
    if (!d_in || !d_out || T < 1 || C < 1 || d_in == d_out) return c->fail(TTC_ERR_ARG, "adjust_shape: bad argument");
    if (!adjust_map(n1, n2, width, height, &am))
        return c->fail(TTC_ERR_ARG, "adjust_shape: an axis is off by an odd amount of 3 or more -- the reference's adjust_shape (job.py:260-310) does "
                                    "not produce the requested size there and process_tile raises");
    const long per = (long)width * height * C;
    hipLaunchKernelGGL(k_adjust_shape, dim3((unsigned)((per + 255) / 256), T), dim3(256), 0, s, d_in, am, width, height, C, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// X, Y: shape of the result; am (may be null = the array has that shape) maps it onto the array as stored [T, am->n1, am->n2, 2]
ttc_status codec_s1_to_db(ttc_ctx* c, const uint16_t* d_u16, int T, int X, int Y, float* d_out, hipStream_t s, const AdjustMap* amp) {
    if (!d_u16 || !d_out || T < 1 || T > 64) return c->fail(TTC_ERR_ARG, "s1_to_db: bad argument (T in [1, 64])");
    const AdjustMap am = amp ? *amp : AdjustMap{X, Y, 0, 0};
    const int per = am.n1 * am.n2 * 2;
    char* ctl = static_cast<char*>(c->scratch_buf("s1_ctl", 4096 + 128 * 256 * 4));
    if (!ctl) return c->fail(TTC_ERR_NOMEM, "s1 scratch");
    S1Sel* sel = reinterpret_cast<S1Sel*>(ctl);                 // 64 x 24 B
    unsigned* hist = reinterpret_cast<unsigned*>(ctl + 4096);   // [T][2][256]
    const int nh = T * 2 * 256;
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(k_s1_clear, dim3((nh + 255) / 256), dim3(256), 0, s, hist, nh);
        hipLaunchKernelGGL(k_s1_hist, dim3(96, T), dim3(256), 0, s, d_u16, per, sel, pass, hist);
        hipLaunchKernelGGL(k_s1_pick, dim3(T, 2), dim3(64), 0, s, sel, per, pass, hist);
    }
    hipLaunchKernelGGL(k_s1_finish, dim3((X * Y * 2 + 255) / 256, T), dim3(256), 0, s, d_u16, sel, per, am, X, Y, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
