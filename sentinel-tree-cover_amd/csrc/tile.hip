// Per-tile numeric core around the model: everything process_subtiles
// (src/download_and_predict_job.py:1125-1483) does with numpy, as HBM-streaming HIP kernels.
//
//   k_tile_temporal   medians over dates (job.py:1152-1160), 0/1 repair with the running median
//                     (job.py:1039-1047), spectral indices (src/preprocessing/indices.py), date
//                     regrid + Whittaker + monthly mean as ONE 12 x T operator
//                     (src/downloading/utils.py:176-347, src/preprocessing/whittaker_smoother.py),
//                     quarterly medians (job.py:1274-1278)              -> planar [L][14][X][Y]
//   k_tile_s1         Sentinel-1 median / quarterly medians (job.py:1174, :1277-1278)
//   k_assemble        window cut + 7-px reflect pad at tile edges + 17-channel assembly +
//                     normalisation (job.py:1355-1407, :316-325) straight into the model's padded
//                     planar frame buffer
//   k_bright_flags / k_bright_dist   identify_bright_bare_surfaces (job.py:1099-1122)
//   k_post            clear-image statistics, no-image mask (job.py:1363-1366, :1451-1472), bright
//                     attenuation and np.around(.,3) (job.py:1480-1482)
//
// Tile arrays arrive in the reference's layout ([T, X, Y, C] float32); intermediates are planar
// [C][X][Y] so that every later access is coalesced along Y.
#include "ttc_internal.h"
#include "h16_common.h"

namespace {

constexpr int kMaxWin = 64;
constexpr int kMaxT = 32;

struct WinDesc {          // one model window, job.py:1295-1317 + tof_downloading.py:498-524
    int sx, sy;           // start of the input slice in the tile
    int lx, ly;           // slice length (size+7 at tile edges, size+14 inside)
    int fx, fy;           // reflect-pad amount in FRONT of the slice (7 on the first row/col, else 0)
    int qx;               // front pad the reference applies to the clear-count map on axis 0 (job.py:1395 quirk)
    int ox, oy;           // output (folder) origin
};
struct WinTable { int n; WinDesc w[kMaxWin]; };
struct WMat { float w[12 * kMaxT]; unsigned keep; int T; int Tk; };
struct Norm { float lo[17], hi[17], mid[17], half[17]; };   // float32-rounded Python-float constants

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// slice-local coordinate (after the 7-px tile-edge reflect pad) -> tile coordinate
__device__ __forceinline__ int win_to_tile(int l, int start, int len, int front) {
    return start + reflect_idx(l - front, len);
}

// ---- float32 formulas evaluated exactly as numpy does (no FMA contraction) -------------------
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

__device__ float idx_evi(float b, float r, float n) {
#pragma clang fp contract(off)
    b = clip01(b); r = clip01(r); n = clip01(n);
    const float den = ((n + (6.0f * r)) - (7.5f * b)) + 1.0f;
    const float e = 2.5f * ((n - r) / den);
    return fminf(fmaxf(e, -1.5f), 1.5f);
}
__device__ float idx_bi(float b2, float b4, float b8, float b11) {
#pragma clang fp contract(off)
    b2 = clip01(b2); b4 = clip01(b4); b8 = clip01(b8); b11 = clip01(b11);
    const float a = b11 + b4, c = b8 + b2;
    const float v = (a - c) / ((a + c) + 1e-5f);
    return fminf(fmaxf(v, -1.0f), 1.0f);
}
__device__ float idx_msavi2(float r, float n) {
#pragma clang fp contract(off)
    r = clip01(r); n = clip01(n);
    const float t = 2.0f * n + 1.0f;
    float s = t * t - 8.0f * (n - r);
    if (s < 0.0f) s = 0.0f;
    const float v = (t - sqrtf(s)) / 2.0f;
    return fminf(fmaxf(v, -1.0f), 1.0f);
}
__device__ float idx_grndvi(float g, float r, float n) {
#pragma clang fp contract(off)
    g = clip01(g); r = clip01(r); n = clip01(n);
    const float gr = g + r;
    return (n - gr) / ((n + gr) + 1e-5f);
}

// ---- median of the first `cnt` members of a TM-array whose non-members are +inf ---------------
template <int TM>
__device__ __forceinline__ void bitonic_sort(float (&a)[TM]) {
#pragma unroll
    for (int k = 2; k <= TM; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float lo = fminf(a[i], a[l]), hi = fmaxf(a[i], a[l]);
                    if ((i & k) == 0) { a[i] = lo; a[l] = hi; } else { a[i] = hi; a[l] = lo; }
                }
            }
        }
    }
}

template <int TM>
__device__ __forceinline__ float median_masked(const float (&v)[TM], unsigned mask, int cnt) {
    float a[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) a[t] = ((mask >> t) & 1u) ? v[t] : INFINITY;
    bitonic_sort<TM>(a);
    const int i0 = (cnt - 1) >> 1, i1 = cnt >> 1;
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) { if (t == i0) lo = a[t]; if (t == i1) hi = a[t]; }
    return (lo + hi) * 0.5f;      // == numpy's mean of the two middle values in float32
}

__device__ __forceinline__ float med3(float a, float b, float c) {
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}

// deal_w_missing_px value repair (job.py:1039-1047): sequentially, each 0 (then each 1) of a kept
// date becomes the median of the CURRENT kept series.
template <int TM>
__device__ __forceinline__ void fix_zero_one(float (&v)[TM], unsigned keep, int Tk) {
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const float bad = pass == 0 ? 0.0f : 1.0f;
        bool any = false;
#pragma unroll
        for (int t = 0; t < TM; ++t) any |= ((keep >> t) & 1u) && v[t] == bad;
        if (!any) continue;
#pragma unroll 1
        for (int t = 0; t < TM; ++t) {              // rolled: one inlined sort per pass, taken rarely
            float vt = 0.f;
#pragma unroll
            for (int u = 0; u < TM; ++u) if (u == t) vt = v[u];
            if (((keep >> t) & 1u) && vt == bad) {
                const float m = median_masked<TM>(v, keep, Tk);
#pragma unroll
                for (int u = 0; u < TM; ++u) if (u == t) v[u] = m;
            }
        }
    }
}

template <int TM>
__device__ __forceinline__ void smooth_store(const float (&v)[TM], const WMat& wm, int L, float* dst, long cstride_L) {
    // dst -> sm[0][ch][p]; frame stride cstride_L
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t) if (t < wm.T) acc = fmaf(wm.w[k * kMaxT + t], v[t], acc);
        m[k] = acc;
    }
    if (L == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * cstride_L] = med3(m[3 * q], m[3 * q + 1], m[3 * q + 2]);
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) dst[k * cstride_L] = m[k];
    }
}

// TWO waves per 64 pixels.  The [T, X, Y, 10] stack is pixel-major (40-byte records): a thread-per-pixel load touches 20
// cache lines per instruction for 512 useful bytes, and the kernel sat at 0.76 TB/s.  The block therefore copies its
// 64 x 10 floats per date into LDS with coalesced 16-byte DMA copies and every lane then picks its pixel's series out of LDS
// (40-byte lane stride: conflict-free for ds_read_b64).  The staged tile (480 B per pixel at T = 12) caps a CU at ~5 blocks, and
// with one wave per block that was ONE wave per SIMD running ~9 k dependent VALU operations per pixel (14 sorting networks,
// 14 12 x T products, the spectral indices twice): 27 % of the VALU rate.  The two waves of a block now split that work over
// the same staged pixels -- wave 0: the medians over all dates (bands, raw indices); wave 1: 0 / 1 repair, the temporal
// operator on bands and on the indices of the repaired bands -- so a SIMD holds two independent instruction streams.
template <int TM>
__global__ __launch_bounds__(128) void k_tile_temporal(const float* __restrict__ s2, const WMat* __restrict__ wmp, int npix, int L,
                                                       float* __restrict__ sm, float* __restrict__ med) {
    const WMat& wm = *wmp;              // device memory (built by the host mirror or by k_build_wmat); uniform indices -> scalar loads
    extern __shared__ __attribute__((aligned(16))) float stage[];      // [T][64 px][10]
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int p0 = blockIdx.x * 64;
    const int T = wm.T;
    {
        const int nfl = min(64, npix - p0) * 10;                      // floats of this block per date
        if (nfl == 640 && (npix & 1) == 0) {     // (odd pixel counts break the 16-byte alignment of odd dates)
            // global -> LDS DMA (wave-uniform LDS base + lane * 16 = exactly this linear copy); no staging registers; the two
            // waves take alternate dates
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                if (t < T && (t & 1) == role) {
                    const float* src = s2 + ((long)t * npix + p0) * 10;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (k < 2 || lane < 32)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (lane + 64 * k) * 4),
                                                             (__attribute__((address_space(3))) void*)(stage + t * 640 + 256 * k), 16, 0, 0);
                    }
                }
            }
        } else {                                                       // ragged last block / unaligned: element-wise
            for (int t = 0; t < T; ++t)
                for (int i = threadIdx.x; i < nfl; i += 128) stage[t * 640 + i] = s2[((long)t * npix + p0) * 10 + i];
        }
    }
    __syncthreads();
    const int p = p0 + lane;
    if (p >= npix) return;
    const unsigned all = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
    const long fstride = 14L * npix;     // sm frame stride
    const float2* px = reinterpret_cast<const float2*>(stage) + lane * 5;
    const int tstride = 320;             // float2 per date in the stage

    // ---- bands 0,1,2,3,8,9 (the index inputs + the last band) ----
    float b0[TM], b1[TM], b2[TM], b3[TM], b8[TM], b9[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < T) {
            const float2 q0 = px[t * tstride], q1 = px[t * tstride + 1], q4 = px[t * tstride + 4];
            b0[t] = q0.x; b1[t] = q0.y; b2[t] = q1.x; b3[t] = q1.y; b8[t] = q4.x; b9[t] = q4.y;
        } else { b0[t] = b1[t] = b2[t] = b3[t] = b8[t] = b9[t] = 0.f; }
    }
    if (role == 0) {
        // medians over ALL dates of the raw bands and of the raw per-date indices (job.py:1152-1160)
        med[0L * npix + p] = median_masked<TM>(b0, all, T);
        med[1L * npix + p] = median_masked<TM>(b1, all, T);
        med[2L * npix + p] = median_masked<TM>(b2, all, T);
        med[3L * npix + p] = median_masked<TM>(b3, all, T);
        med[8L * npix + p] = median_masked<TM>(b8, all, T);
        med[9L * npix + p] = median_masked<TM>(b9, all, T);
        float ix[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_evi(b0[t], b2[t], b3[t]);
        med[10L * npix + p] = median_masked<TM>(ix, all, T);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_bi(b0[t], b2[t], b3[t], b8[t]);
        med[11L * npix + p] = median_masked<TM>(ix, all, T);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_msavi2(b2[t], b3[t]);
        med[12L * npix + p] = median_masked<TM>(ix, all, T);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_grndvi(b1[t], b2[t], b3[t]);
        med[13L * npix + p] = median_masked<TM>(ix, all, T);
    } else {
        // repair, then smooth bands and the indices of the REPAIRED bands (job.py:1067-1082)
        fix_zero_one<TM>(b0, wm.keep, wm.Tk); fix_zero_one<TM>(b1, wm.keep, wm.Tk); fix_zero_one<TM>(b2, wm.keep, wm.Tk);
        fix_zero_one<TM>(b3, wm.keep, wm.Tk); fix_zero_one<TM>(b8, wm.keep, wm.Tk); fix_zero_one<TM>(b9, wm.keep, wm.Tk);
        smooth_store<TM>(b0, wm, L, sm + 0L * npix + p, fstride);
        smooth_store<TM>(b1, wm, L, sm + 1L * npix + p, fstride);
        smooth_store<TM>(b2, wm, L, sm + 2L * npix + p, fstride);
        smooth_store<TM>(b3, wm, L, sm + 3L * npix + p, fstride);
        smooth_store<TM>(b8, wm, L, sm + 8L * npix + p, fstride);
        smooth_store<TM>(b9, wm, L, sm + 9L * npix + p, fstride);
        float ix[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_evi(b0[t], b2[t], b3[t]);
        smooth_store<TM>(ix, wm, L, sm + 10L * npix + p, fstride);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_bi(b0[t], b2[t], b3[t], b8[t]);
        smooth_store<TM>(ix, wm, L, sm + 11L * npix + p, fstride);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_msavi2(b2[t], b3[t]);
        smooth_store<TM>(ix, wm, L, sm + 12L * npix + p, fstride);
#pragma unroll
        for (int t = 0; t < TM; ++t) ix[t] = idx_grndvi(b1[t], b2[t], b3[t]);
        smooth_store<TM>(ix, wm, L, sm + 13L * npix + p, fstride);
    }
    // ---- bands 4,5,6,7 ----
    {
        float c4[TM], c5[TM], c6[TM], c7[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t < T) {
                const float2 q2 = px[t * tstride + 2], q3 = px[t * tstride + 3];
                c4[t] = q2.x; c5[t] = q2.y; c6[t] = q3.x; c7[t] = q3.y;
            } else { c4[t] = c5[t] = c6[t] = c7[t] = 0.f; }
        }
        if (role == 0) {
            med[4L * npix + p] = median_masked<TM>(c4, all, T);
            med[5L * npix + p] = median_masked<TM>(c5, all, T);
            med[6L * npix + p] = median_masked<TM>(c6, all, T);
            med[7L * npix + p] = median_masked<TM>(c7, all, T);
        } else {
            fix_zero_one<TM>(c4, wm.keep, wm.Tk); fix_zero_one<TM>(c5, wm.keep, wm.Tk);
            fix_zero_one<TM>(c6, wm.keep, wm.Tk); fix_zero_one<TM>(c7, wm.keep, wm.Tk);
            smooth_store<TM>(c4, wm, L, sm + 4L * npix + p, fstride);
            smooth_store<TM>(c5, wm, L, sm + 5L * npix + p, fstride);
            smooth_store<TM>(c6, wm, L, sm + 6L * npix + p, fstride);
            smooth_store<TM>(c7, wm, L, sm + 7L * npix + p, fstride);
        }
    }
}

// Sentinel-1: [12][X][Y][2] -> median over 12 (job.py:1174) and L-step series (job.py:1277-1278)
__global__ void k_tile_s1(const float* __restrict__ s1, int npix, int L, float* __restrict__ s1q,
                          float* __restrict__ s1med) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if (t < 12) {
            const float2 q = reinterpret_cast<const float2*>(s1)[(long)t * npix + p];
            a[t] = q.x; b[t] = q.y;
        } else { a[t] = b[t] = 0.f; }
    }
    s1med[p] = median_masked<16>(a, 0xfffu, 12);
    s1med[npix + p] = median_masked<16>(b, 0xfffu, 12);
    if (L == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s1q[(2L * q) * npix + p] = med3(a[3 * q], a[3 * q + 1], a[3 * q + 2]);
            s1q[(2L * q + 1) * npix + p] = med3(b[3 * q], b[3 * q + 1], b[3 * q + 2]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 12; ++t) { s1q[(2L * t) * npix + p] = a[t]; s1q[(2L * t + 1) * npix + p] = b[t]; }
    }
}

// window assembly straight into the model's padded frame buffer: fp32 planar [win][L+1][17][PP] (BF < 0), or -- for the 16-bit
// conv engine -- its channel-blocked hi / lo pairs [win][L+1][3][PP][8] (BF = 0 fp16, 1 bf16; channels 17..23 zero), which saves
// the planar round trip and the conversion pass
template <int BF>
__global__ __launch_bounds__(256) void k_assemble(const float* __restrict__ sm, const float* __restrict__ med,
                                                  const float* __restrict__ s1q, const float* __restrict__ s1med,
                                                  const float* __restrict__ dem, WinTable wt, Norm nm, int X, int Y,
                                                  int W, int L, float* __restrict__ frames, uint4* __restrict__ fhi, uint4* __restrict__ flo) {
#pragma clang fp contract(off)
    const int Wp = W + 2, PP = Wp * Wp;
    const int f = blockIdx.y, wi = blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, pxx = p - py * Wp;
    int lx = py - 1, ly = pxx - 1;
    const bool border = lx < 0 || lx >= W || ly < 0 || ly >= W;
    const bool last = (f == L);
    const long img = (long)wi * (L + 1) + f;
    float* dst = BF < 0 ? frames + (img * 17) * PP + p : nullptr;
    float o[24];
#pragma unroll
    for (int c = 0; c < 24; ++c) o[c] = 0.0f;
    if (!(last && border)) {
        lx = reflect_idx(lx, W); ly = reflect_idx(ly, W);        // ConvGRU reflect pad (model.py:250)
        const WinDesc& w = wt.w[wi];
        const int tx = win_to_tile(lx, w.sx, w.lx, w.fx), ty = win_to_tile(ly, w.sy, w.ly, w.fy);
        const long npix = (long)X * Y, tp = (long)tx * Y + ty;
#pragma unroll
        for (int c = 0; c < 17; ++c) {
            float v;
            if (c == 10) v = dem[tp];
            else if (c == 11 || c == 12) v = last ? s1med[(c - 11) * npix + tp] : s1q[(2L * f + (c - 11)) * npix + tp];
            else {
                const int ch = c < 10 ? c : c - 3;               // 13..16 -> smoothed / median index 10..13
                v = last ? med[ch * npix + tp] : sm[((long)f * 14 + ch) * npix + tp];
            }
            v = fminf(fmaxf(v, nm.lo[c]), nm.hi[c]);             // normalize_subtile, job.py:316-325 (float32)
            o[c] = (v - nm.mid[c]) / nm.half[c];
        }
    }
    if constexpr (BF < 0) {
#pragma unroll
        for (int c = 0; c < 17; ++c) dst[(long)c * PP] = o[c];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = o[8 * k + j];
            b16_store8<BF>(fhi, flo, (img * 3 + k) * PP + p, v8);
        }
    }
}

// identify_bright_bare_surfaces, part 1 (job.py:1110-1114): per window pixel, more than one frame
// with NIR/(SWIR+0.01) < 0.9, mean(B,G,R) > 0.2 and EVI < 0.3 (on the un-normalised window)
__global__ void k_bright_flags(const float* __restrict__ sm, const float* __restrict__ med, WinTable wt, int X, int Y,
                               int W, int L, unsigned char* __restrict__ flags) {
#pragma clang fp contract(off)
    const int wi = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= W * W) return;
    const int lx = p / W, ly = p - lx * W;
    const WinDesc& w = wt.w[wi];
    const int tx = win_to_tile(lx, w.sx, w.lx, w.fx), ty = win_to_tile(ly, w.sy, w.ly, w.fy);
    const long npix = (long)X * Y, tp = (long)tx * Y + ty;
    int cnt = 0;
    for (int f = 0; f <= L; ++f) {
        const float* src = f == L ? med : sm + (long)f * 14 * npix;
        const float b = src[0 * npix + tp], g = src[1 * npix + tp], r = src[2 * npix + tp], n = src[3 * npix + tp],
                    sw = src[8 * npix + tp];
        const bool c1 = (n / (sw + 0.01f)) < 0.9f;
        const bool c2 = (((b + g) + r) / 3.0f) > 0.2f;
        const bool c3 = idx_evi(b, r, n) < 0.3f;
        cnt += (c1 && c2 && c3) ? 1 : 0;
    }
    flags[(long)wi * W * W + p] = cnt > 1 ? 1 : 0;
}

// part 2 (job.py:1115-1122): erode(diamond 2) -> dilate(diamond 1) -> squared distance to the nearest
// remaining bright pixel, capped at 9, on the [7:-7] crop.  One workgroup per window, bitmaps in LDS.
__global__ __launch_bounds__(1024) void k_bright_dist(const unsigned char* __restrict__ flags, int W, int size,
                                                      unsigned char* __restrict__ d2out) {
    extern __shared__ unsigned char lds[];
    unsigned char* A = lds; unsigned char* E = lds + W * W; unsigned char* B = lds + 2 * W * W;
    const int wi = blockIdx.x, P = W * W;
    for (int p = threadIdx.x; p < P; p += blockDim.x) A[p] = flags[(long)wi * P + p];
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const int x = p / W, y = p - x * W;
        bool all = true;
        for (int dx = -2; dx <= 2; ++dx)
            for (int dy = -(2 - abs(dx)); dy <= 2 - abs(dx); ++dy) {
                const int xx = x + dx, yy = y + dy;
                if (xx >= 0 && xx < W && yy >= 0 && yy < W) all &= A[xx * W + yy] != 0;
            }
        E[p] = all;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const int x = p / W, y = p - x * W;
        bool any = E[p];
        if (x > 0) any |= E[p - W] != 0;
        if (x < W - 1) any |= E[p + W] != 0;
        if (y > 0) any |= E[p - 1] != 0;
        if (y < W - 1) any |= E[p + 1] != 0;
        B[p] = any;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < size * size; o += blockDim.x) {
        const int x = o / size + 7, y = o % size + 7;
        int best = 9;
        for (int dx = -3; dx <= 3; ++dx)
            for (int dy = -3; dy <= 3; ++dy) {
                const int xx = x + dx, yy = y + dy, d = dx * dx + dy * dy;
                if (d < best && xx >= 0 && xx < W && yy >= 0 && yy < W && B[xx * W + yy]) best = d;
            }
        d2out[(long)wi * size * size + o] = (unsigned char)best;
    }
}

// post-masks + rounding; one workgroup per window
// number of kept dates whose interpolation weight is < 0.33 (job.py:1355-1362), once per tile pixel: every pixel is
// visited by up to four windows and twice per window, and k_post runs on 36 workgroups only
__global__ void k_clear_map(const float* __restrict__ interp, const WMat* __restrict__ wmp, int T, int npix, unsigned char* __restrict__ cc) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const unsigned keep = wmp->keep;
    int c = 0;
    for (int t = 0; t < T; ++t)
        if ((keep >> t) & 1u) c += interp[(long)t * npix + p] < 0.33f ? 1 : 0;
    cc[p] = (unsigned char)c;
}
struct PostArgs {
    const float* probs; const unsigned char* cc; const unsigned char* d2;
    float* out; float* out_raw;
    WinTable wt; const WMat* wm; int T, X, Y, size, n_dates_ok;      // n_dates_ok < 0: the kept-date count of *wm
};

__device__ __forceinline__ int clear_count(const PostArgs& a, int tx, int ty) { return a.cc[(long)tx * a.Y + ty]; }

__global__ __launch_bounds__(1024) void k_post(PostArgs a) {
    extern __shared__ unsigned char lds[];
    __shared__ int s_z, s_z1, s_blk[81];
    const int wi = blockIdx.x, size = a.size, m = size + 2;
    const WinDesc& w = a.wt.w[wi];
    unsigned char* M0 = lds; unsigned char* M1 = lds + m * m;
    if (threadIdx.x == 0) { s_z = 0; s_z1 = 0; }
    if (threadIdx.x < 81) s_blk[threadIdx.x] = 0;
    __syncthreads();
    // window-level "no clear image" test: np.percentile(min_clear, 50) < 1 on the unpadded slice (job.py:1363-1366)
    {
        int z = 0, z1 = 0;
        for (int i = threadIdx.x; i < w.lx * w.ly; i += blockDim.x) {
            const int c = clear_count(a, w.sx + i / w.ly, w.sy + i % w.ly);
            z += c == 0; z1 += c <= 1;
        }
        for (int k = 32; k >= 1; k >>= 1) { z += __shfl_xor(z, k); z1 += __shfl_xor(z1, k); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&s_z, z); atomicAdd(&s_z1, z1); }
    }
    const bool has_mask = (size == 158 || size == 142);
    const int nb = size == 158 ? 4 : 9, bs = size == 158 ? 40 : 16;
    if (has_mask) {
        // clear map on the [6:-6] crop of the (size+14)^2 padded count map, with the reference's pad sides
        for (int i = threadIdx.x; i < m * m; i += blockDim.x) {
            const int lx = i / m + 6, ly = i % m + 6;
            const int tx = win_to_tile(lx, w.sx, w.lx, w.lx == size + 7 ? w.qx : 0);
            const int ty = win_to_tile(ly, w.sy, w.ly, w.fy);
            M0[i] = clear_count(a, tx, ty) >= 1;
        }
        __syncthreads();
        // 8-connected dilation x6 == 13x13 max, separable; twice (clear -> !dilated -> dilated again)
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = threadIdx.x; i < m * m; i += blockDim.x) {
                const int x = i / m, y = i % m;
                bool v = false;
                for (int d = -6; d <= 6; ++d) { const int yy = y + d; if (yy >= 0 && yy < m) v |= M0[x * m + yy] != 0; }
                M1[i] = v;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < m * m; i += blockDim.x) {
                const int x = i / m, y = i % m;
                bool v = false;
                for (int d = -6; d <= 6; ++d) { const int xx = x + d; if (xx >= 0 && xx < m) v |= M1[xx * m + y] != 0; }
                M0[i] = rep == 0 ? !v : v;
            }
            __syncthreads();
        }
        // block vote (job.py:1459-1463 / :1467-1471)
        for (int i = threadIdx.x; i < m * m; i += blockDim.x)
            if (M0[i]) atomicAdd(&s_blk[(i / m / bs) * nb + (i % m) / bs], 1);
    }
    __syncthreads();
    const int n = w.lx * w.ly, k = n / 2;
    bool no_images = (n & 1) ? (s_z >= k + 1) : (s_z >= k && s_z1 >= k + 1);
    if ((a.n_dates_ok < 0 ? a.wm->Tk : a.n_dates_ok) < 2) no_images = true;
    const int thr = size == 158 ? 400 : 192;
    for (int o = threadIdx.x; o < size * size; o += blockDim.x) {
        const long gi = (long)wi * size * size + o;
        float p = no_images ? 255.0f : a.probs[gi];
        if (a.out_raw) a.out_raw[gi] = p;
        if (has_mask) {
            const int li = o / size + 1, lj = o % size + 1;
            if (s_blk[(li / bs) * nb + lj / bs] > thr) p = 255.0f;
        }
        // preds * bright_surface in float64, np.around(., 3), astype(float32)  (job.py:1480-1482)
        const double blur = sqrt((double)a.d2[gi]) / 3.0;
        a.out[gi] = (float)(rint(((double)p * blur) * 1000.0) / 1000.0);
    }
}

// id_missing_px (src/preprocessing/interpolation.py:5-23): per date, pixels with more than one band == 0 or >= 1
__global__ void k_missing_counts(const float* __restrict__ s2, int npix, int* __restrict__ counts) {
    const int t = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const float2* v = reinterpret_cast<const float2*>(s2 + ((long)t * npix + p) * 10);      // 8-byte loads: half the load instructions
        int bad = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) { const float2 u = v[b]; bad += (u.x == 0.0f) + (u.x >= 1.0f) + (u.y == 0.0f) + (u.y >= 1.0f); }
        c += bad > 1;
    }
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[t], c);
}

// interpolate_na_vals (interpolation.py:42-56): any NaN in a pixel-band series makes bn.median NaN -> 0,
// so every NaN of that series becomes 0;  optional 0/1 repair (job.py:1039-1047) in place.
// interpolate_na_vals' NaN -> 0 (interpolation.py:42-56 as tile_fix_missing(do_nan) applies it: element-wise) and id_missing_px's per-date
// counts (job.py:1032) in ONE pass over the stack (round 6: the single-call path ran k_fix_missing and k_missing_counts back to back, 2 x 183 MB
// fetched to rewrite the rare NaN).  A record is re-written only where a NaN was found.
__global__ void k_nanfix_counts(float* __restrict__ s2, int npix, int* __restrict__ counts) {
    const int t = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        float2* v = reinterpret_cast<float2*>(s2 + ((long)t * npix + p) * 10);
        int bad = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            float2 u = v[b];
            const bool fx = isnan(u.x), fy = isnan(u.y);
            if (fx | fy) { if (fx) u.x = 0.0f; if (fy) u.y = 0.0f; v[b] = u; }
            bad += (u.x == 0.0f) + (u.x >= 1.0f) + (u.y == 0.0f) + (u.y >= 1.0f);
        }
        c += bad > 1;
    }
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[t], c);
}
template <int TM>
__global__ void k_fix_missing(float* __restrict__ s2, int T, int npix, int do_nan, int do_zero_one) {
    // thread = the float2 of two neighbouring bands of one pixel (round 5: 8-byte loads / stores; one band per thread before)
    const long i2 = (long)blockIdx.x * blockDim.x + threadIdx.x;     // pixel*5 + band pair
    if (i2 >= (long)npix * 5) return;
    float2* base = reinterpret_cast<float2*>(s2) + i2;
    const long tstride = (long)npix * 5;
    float v[2][TM];
    bool anynan[2] = {false, false};
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const float2 u = t < T ? base[(long)t * tstride] : float2{0.f, 0.f};
        v[0][t] = u.x; v[1][t] = u.y;
        anynan[0] |= (t < T) && isnan(u.x); anynan[1] |= (t < T) && isnan(u.y);
    }
    bool changed = false;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (do_nan && anynan[e]) {
#pragma unroll
            for (int t = 0; t < TM; ++t) if (t < T && isnan(v[e][t])) v[e][t] = 0.0f;
            changed = true;
        }
        if (do_zero_one) {
            const unsigned all = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
            bool any = false;
#pragma unroll
            for (int t = 0; t < TM; ++t) any |= t < T && (v[e][t] == 0.0f || v[e][t] == 1.0f);
            if (any) { fix_zero_one<TM>(v[e], all, T); changed = true; }
        }
    }
    if (changed)
#pragma unroll
        for (int t = 0; t < TM; ++t) if (t < T) base[(long)t * tstride] = float2{v[0][t], v[1][t]};
}

// window table: job.py:1295-1317 + make_overlapping_windows (src/tof/tof_downloading.py:498-524)
bool build_windows(int X, int Y, int size, WinTable& wt) {
    const int n_rows = 6, diff = 7;
    if (X < size + 2 * diff || Y < size + 2 * diff) return false;
    const int gx = (X - size + n_rows - 2) / (n_rows - 1), gy = (Y - size + n_rows - 2) / (n_rows - 1);
    std::vector<int> xs, ys;
    for (int v = 0; v < X - size; v += gx) xs.push_back(v);
    xs.push_back(X - size);
    for (int v = 0; v < Y - size; v += gy) ys.push_back(v);
    ys.push_back(Y - size);
    if (xs.size() * ys.size() > kMaxWin) return false;
    wt.n = 0;
    for (size_t ix = 0; ix < xs.size(); ++ix)
        for (size_t iy = 0; iy < ys.size(); ++iy) {
            WinDesc d{};
            const bool fx0 = ix == 0, fxl = ix + 1 == xs.size(), fy0 = iy == 0, fyl = iy + 1 == ys.size();
            d.ox = xs[ix]; d.oy = ys[iy];
            d.sx = fx0 ? 0 : xs[ix] - diff; d.sy = std::max(0, ys[iy] - diff);
            d.lx = size + ((fx0 || fxl) ? diff : 2 * diff);
            d.ly = size + ((fy0 || fyl) ? diff : 2 * diff);
            d.fx = fx0 ? diff : 0; d.fy = fy0 ? diff : 0;
            d.qx = fyl ? 0 : diff;          // job.py:1395 pads axis 0 with the (stale) y-axis amounts
            wt.w[wt.n++] = d;
        }
    return true;
}

}  // namespace


// Border strip (src/resegment_tiles_wide.py:997-1000, :1058-1066): regularize_and_smooth (:772-790) on the 10 bands and
// make_and_smooth_indices (job.py:1009-1028) on the indices of the same dates, both = the 12 x T operator of
// k_tile_temporal, kept as 12 monthly steps in one [12][npix][14] array (bands | indices) that the super-resolution and
// ttc_border_subtiles then work on.  One thread per pixel; the strip is ~0.4 Mpx, this is not a hot kernel.
__global__ __launch_bounds__(256) void k_strip_smooth(const float* __restrict__ s2, WMat wm, int npix, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int T = wm.T;
    float acc[12][14];
#pragma unroll
    for (int k = 0; k < 12; ++k)
#pragma unroll
        for (int c = 0; c < 14; ++c) acc[k][c] = 0.f;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        float v[14];
        const float* src = s2 + ((long)t * npix + p) * 10;
#pragma unroll
        for (int c = 0; c < 10; ++c) v[c] = src[c];
        v[10] = idx_evi(v[0], v[2], v[3]);
        v[11] = idx_bi(v[0], v[2], v[3], v[8]);
        v[12] = idx_msavi2(v[2], v[3]);
        v[13] = idx_grndvi(v[1], v[2], v[3]);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float w = wm.w[k * kMaxT + t];
#pragma unroll
            for (int c = 0; c < 14; ++c) acc[k][c] = fmaf(w, v[c], acc[k][c]);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float* dst = out + ((long)k * npix + p) * 14;
#pragma unroll
        for (int c = 0; c < 14; ++c) dst[c] = acc[k][c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The 12 x T temporal operator W = P (I + 100 D2'D2)^-1 R(dates) built ON THE DEVICE from the per-date missing-pixel counts
// (deal_w_missing_px's screening, job.py:1031-1037: a date survives when fewer than X^2 / 10 pixels are missing) and the
// acquisition days, so that the single-call tile path needs no host round trip.  Same arithmetic as the host mirror
// sentinel-tree-cover_amd/temporal.py (calculate_and_save_best_images, src/downloading/utils.py:176-347: per 15-day grid step the
// <= 2 images before and <= 2 after, distance weights, wrap-around / mirroring at the year ends; duplicate dates make the
// reference raise -> all-zero operator, job.py:1073-1080).  One workgroup: thread r < 24 builds row r of R, then thread
// k < 12 row k of W.  minv: the constant 12 x 24 matrix P (I + 100 D2'D2)^-1 in double.
__global__ void k_build_wmat(const int* __restrict__ counts, int thr, const int* __restrict__ dates, int T,
                             const double* __restrict__ minv, WMat* __restrict__ out, int* __restrict__ status) {
    __shared__ float R[24][kMaxT];
    __shared__ int dd[kMaxT], pos[kMaxT], nk, bad;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int n = 0;
        unsigned keep = 0;
        for (int t = 0; t < T; ++t)
            if (counts[t] < thr) {
                int d = dates[t];
                if (d < -100) d = ((d % 365) + 365) % 365;                  // utils.py:190 (Python modulo)
                dd[n] = d; pos[n] = t; keep |= 1u << t; ++n;
            }
        nk = n; bad = 0;
        out->keep = keep; out->T = T; out->Tk = n;
        if (status) status[1] = n;
    }
    for (int i = tid; i < 24 * kMaxT; i += blockDim.x) (&R[0][0])[i] = 0.f;
    __syncthreads();
    const int n = nk;
    if (tid < 24 && n > 0) {
        const int g = 15 * tid;
        int dmin = dd[0], dmax = dd[0];
        for (int i = 1; i < n; ++i) { dmin = min(dmin, dd[i]); dmax = max(dmax, dd[i]); }
        // prior = d[d < 5][-2:], after = d[d >= -5][:2]   (d = dates - g, array order)
        int P[2], A[2], np_ = 0, na = 0;
        for (int i = 0; i < n; ++i) {
            const int d = dd[i] - g;
            if (d < 5) { if (np_ < 2) P[np_++] = d; else { P[0] = P[1]; P[1] = d; } }
        }
        for (int i = 0; i < n && na < 2; ++i) { const int d = dd[i] - g; if (d >= -5) A[na++] = d; }
        if (np_ > 0) {                                                       // keep those within 100 days of the nearest
            int mx = P[0]; for (int j = 1; j < np_; ++j) mx = max(mx, P[j]);
            int m = 0; for (int j = 0; j < np_; ++j) if (P[j] > mx - 100) P[m++] = P[j];
            np_ = m;
        }
        if (na > 0) {
            int mn = A[0]; for (int j = 1; j < na; ++j) mn = min(mn, A[j]);
            int m = 0; for (int j = 0; j < na; ++j) if (A[j] < mn + 100) A[m++] = A[j];
            na = m;
        }
        int p_shift = 0, a_shift = 0;
        if (np_ == 0) {
            if (dmin >= 90) { P[0] = dd[n - 1] - g; np_ = 1; p_shift = 365; }
            else { for (int j = 0; j < na; ++j) P[j] = A[j]; np_ = na; }
        }
        if (na == 0) {
            if (dmax <= 270) { A[0] = dd[0] - g; na = 1; a_shift = 365; }
            else { for (int j = 0; j < np_; ++j) A[j] = P[j]; na = np_; }
        }
        double pd[2], ad[2], pw[2], aw[2];
        for (int j = 0; j < np_; ++j) pd[j] = fmax(fabs((double)(P[j] - p_shift)), 1.0);
        for (int j = 0; j < na; ++j) ad[j] = fmax(fabs((double)(A[j] + a_shift)), 1.0);
        const double closest = fmax(pd[np_ - 1] + ad[0], 2.0);
        for (int j = 0; j < np_; ++j) pw[j] = fabs(1.0 - pd[j] / closest);
        if (np_ == 2) pw[0] = fabs((pd[1] / pd[0]) * pw[1]);                 // prior: far .. near
        for (int j = 0; j < na; ++j) aw[j] = fabs(1.0 - ad[j] / closest);
        if (na == 2) aw[1] = fabs((ad[0] / ad[1]) * aw[0]);                  // after: near .. far
        double total = 0.0;
        for (int j = 0; j < np_; ++j) total += pw[j];
        for (int j = 0; j < na; ++j) total += aw[j];
        // indices: np.flatnonzero(np.isin(dates, g + prior))[:2], np.flatnonzero(np.isin(dates, g + after))[-2:]
        int pi[2], ai[2], npi = 0, nai = 0, cnt_a = 0;
        for (int i = 0; i < n; ++i) {
            bool inp = false, ina = false;
            for (int j = 0; j < np_; ++j) inp |= (dd[i] == g + P[j]);
            for (int j = 0; j < na; ++j) ina |= (dd[i] == g + A[j]);
            if (inp && npi < 2) pi[npi++] = i;
            if (ina) { if (nai < 2) ai[nai++] = i; else { ai[0] = ai[1]; ai[1] = i; } ++cnt_a; }
        }
        (void)cnt_a;
        if (npi != np_ || nai != na) bad = 1;                                // duplicate dates: the reference raises
        else {
            for (int j = 0; j < np_; ++j) R[tid][pi[j]] += (float)(pw[j] / total);
            for (int j = 0; j < na; ++j) R[tid][ai[j]] += (float)(aw[j] / total);
        }
    }
    __syncthreads();
    if (tid < 12) {
        for (int t = 0; t < kMaxT; ++t) out->w[tid * kMaxT + t] = 0.f;
        if (!bad)
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int r = 0; r < 24; ++r) acc += minv[tid * 24 + r] * (double)R[r][i];
                out->w[tid * kMaxT + pos[i]] = (float)acc;
            }
    }
}

// P (I + 100 D2'D2)^-1, 12 x 24, in double on the host (whittaker_smoother.py:25-36 + the pair means of :64-67)
static void whittaker_monthly_matrix(double (&M)[12][24]) {
    const int n = 24;
    double A[24][48];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 2 * n; ++j) A[i][j] = (j == i || j == n + i) ? 1.0 : 0.0;
    for (int r = 0; r < n - 2; ++r) {                                        // + 100 * D2' D2, D2 rows [1, -2, 1]
        const double d[3] = {1.0, -2.0, 1.0};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) A[r + a][r + b] += 100.0 * d[a] * d[b];
    }
    for (int col = 0; col < n; ++col) {                                      // Gauss-Jordan with partial pivoting
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
        if (piv != col) for (int j = 0; j < 2 * n; ++j) std::swap(A[piv][j], A[col][j]);
        const double inv = 1.0 / A[col][col];
        for (int j = 0; j < 2 * n; ++j) A[col][j] *= inv;
        for (int r = 0; r < n; ++r)
            if (r != col && A[r][col] != 0.0) {
                const double f = A[r][col];
                for (int j = 0; j < 2 * n; ++j) A[r][j] -= f * A[col][j];
            }
    }
    for (int k = 0; k < 12; ++k)
        for (int j = 0; j < n; ++j) M[k][j] = 0.5 * (A[2 * k][n + j] + A[2 * k + 1][n + j]);
}

#define LAUNCH_T(kern, T, ...)                                                        \
    do {                                                                              \
        if ((T) <= 8) hipLaunchKernelGGL((kern<8>), __VA_ARGS__);                     \
        else if ((T) <= 16) hipLaunchKernelGGL((kern<16>), __VA_ARGS__);              \
        else hipLaunchKernelGGL((kern<32>), __VA_ARGS__);                             \
    } while (0)

ttc_status tile_smooth_strip(ttc_ctx* c, const float* d_s2, int T, int X, int Y, const float* h_wmat, float* d_out, hipStream_t s) {
    if (!d_s2 || !h_wmat || !d_out || X < 1 || Y < 1) return c->fail(TTC_ERR_ARG, "smooth_strip: bad argument");
    if (T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "smooth_strip: T must be in [1, 32]");
    WMat wm{};
    wm.T = T; wm.keep = T >= 32 ? 0xffffffffu : ((1u << T) - 1u); wm.Tk = T;
    for (int k = 0; k < 12; ++k)
        for (int t = 0; t < T; ++t) wm.w[k * kMaxT + t] = h_wmat[k * T + t];
    const long npix = (long)X * Y;
    KTimer kt(c, "strip_smooth", s);
    hipLaunchKernelGGL(k_strip_smooth, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, d_s2, wm, (int)npix, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status tile_missing_counts(ttc_ctx* c, const float* d_s2, int T, int X, int Y, int32_t* d_counts, hipStream_t s) {
    if (!d_s2 || !d_counts || T < 1) return c->fail(TTC_ERR_ARG, "tile_missing_counts: bad argument");
    TTC_HIP(c, hipMemsetAsync(d_counts, 0, sizeof(int32_t) * T, s));
    KTimer kt(c, "missing_counts", s);
    hipLaunchKernelGGL(k_missing_counts, dim3(128, T), dim3(256), 0, s, d_s2, X * Y, d_counts);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status tile_fix_missing(ttc_ctx* c, float* d_s2, int T, int X, int Y, int do_nan, int do_zero_one, hipStream_t s) {
    if (!d_s2 || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "tile_fix_missing: T must be in [1, 32]");
    KTimer kt(c, "fix_missing", s);
    const long n = (long)X * Y * 10;
    LAUNCH_T(k_fix_missing, T, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, d_s2, T, X * Y, do_nan, do_zero_one);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// core of process_subtiles with the temporal operator already in device memory (d_wm); n_dates_ok < 0: taken from *d_wm.
// stop_after_inputs: only the preprocessing half (temporal stage, window assembly + normalisation into the model's input frames)
static ttc_status tile_core(ttc_ctx* c, const float* d_s2, int T, int X, int Y, const WMat* d_wm, const float* d_interp,
                            const float* d_s1, const float* d_dem, const double* h_min, const double* h_max, int size, int n_dates_ok,
                            float* d_windows, float* d_windows_raw, bool stop_after_inputs, hipStream_t s) {
    const int W = c->cfg.win_in, L = c->cfg.length;
    if (size != W - 14) return c->fail(TTC_ERR_ARG, "process_subtiles: size must equal win_in - 14");
    if (c->cfg.win_rows != 0 && c->cfg.win_rows != W) return c->fail(TTC_ERR_ARG, "process_subtiles: needs square windows (win_rows = 0)");
    if (L != 4 && L != 12) return c->fail(TTC_ERR_ARG, "process_subtiles: length must be 4 or 12");
    WinTable wt{};
    if (!build_windows(X, Y, size, wt)) return c->fail(TTC_ERR_ARG, "process_subtiles: tile too small for a 6x6 window grid");
    if (wt.n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "process_subtiles: max_windows < windows per tile");
    const long npix = (long)X * Y;
    float* sm = static_cast<float*>(c->scratch_buf("sm", sizeof(float) * L * 14 * npix));
    float* med = static_cast<float*>(c->scratch_buf("med", sizeof(float) * 14 * npix));
    float* s1q = static_cast<float*>(c->scratch_buf("s1q", sizeof(float) * L * 2 * npix));
    float* s1med = static_cast<float*>(c->scratch_buf("s1med", sizeof(float) * 2 * npix));
    unsigned char* flags = static_cast<unsigned char*>(c->scratch_buf("flags", (size_t)wt.n * W * W));
    unsigned char* d2 = static_cast<unsigned char*>(c->scratch_buf("d2", (size_t)wt.n * size * size));
    float* probs = static_cast<float*>(c->scratch_buf("probs", sizeof(float) * wt.n * size * size));
    if (!sm || !med || !s1q || !s1med || !flags || !d2 || !probs) return c->fail(TTC_ERR_NOMEM, "tile scratch");
    c->named["tile_sm"] = {sm, (size_t)L * 14 * npix};
    c->named["tile_med"] = {med, (size_t)14 * npix};
    c->named["tile_s1q"] = {s1q, (size_t)L * 2 * npix};
    c->named["tile_s1med"] = {s1med, (size_t)2 * npix};
    c->named["tile_probs"] = {probs, (size_t)wt.n * size * size};
    Norm nm{};
    for (int i = 0; i < 17; ++i) {
        nm.lo[i] = (float)h_min[i]; nm.hi[i] = (float)h_max[i];
        nm.mid[i] = (float)((h_max[i] + h_min[i]) / 2.0);
        nm.half[i] = (float)((h_max[i] - h_min[i]) / 2.0);
    }
    const unsigned gp = (unsigned)((npix + 255) / 256);
    { KTimer kt(c, "tile_temporal", s);
      const int TM = T <= 8 ? 8 : (T <= 16 ? 16 : 32);
      if (TM == 32) {
          static LdsConfig cfg32;
          TTC_HIP(c, cfg32.ensure(&k_tile_temporal<32>, (size_t)32 * 640 * 4));
      }
      LAUNCH_T(k_tile_temporal, T, dim3((unsigned)((npix + 63) / 64)), dim3(128), (size_t)std::min(T, TM) * 640 * sizeof(float), s, d_s2, d_wm, (int)npix, L, sm, med);
      TTC_HIP(c, hipGetLastError()); }
    { KTimer kt(c, "tile_s1", s);
      hipLaunchKernelGGL(k_tile_s1, dim3(gp), dim3(256), 0, s, d_s1, (int)npix, L, s1q, s1med);
      TTC_HIP(c, hipGetLastError()); }
    const int PP = (W + 2) * (W + 2);
    bool blocked_frames = false;
    { KTimer kt(c, "assemble", s);
      const dim3 ag((PP + 255) / 256, L + 1, wt.n);
      // 16-bit engine: the blocked pairs directly, unless somebody wants to see the fp32 frames (model feed output, debug keep)
      const bool blocked = c->half() && !c->want_planar_frames && !c->keep_debug && (c->cfg.win_rows == 0 || c->cfg.win_rows == c->cfg.win_in) && c->cfg.n_bands == 17;
      if (!blocked) hipLaunchKernelGGL(k_assemble<-1>, ag, dim3(256), 0, s, sm, med, s1q, s1med, d_dem, wt, nm, X, Y, W, L, c->frames, nullptr, nullptr);
      else if (c->blk_mode() == 1) hipLaunchKernelGGL(k_assemble<1>, ag, dim3(256), 0, s, sm, med, s1q, s1med, d_dem, wt, nm, X, Y, W, L, nullptr,
                                                      c->frames16.hi, c->frames16.lo);
      else hipLaunchKernelGGL(k_assemble<0>, ag, dim3(256), 0, s, sm, med, s1q, s1med, d_dem, wt, nm, X, Y, W, L, nullptr, c->frames16.hi, c->frames16.lo);
      c->frames_planar_valid = !blocked;
      blocked_frames = blocked;
      TTC_HIP(c, hipGetLastError()); }
    if (stop_after_inputs) return TTC_OK;
    { KTimer kt(c, "bright", s);
      hipLaunchKernelGGL(k_bright_flags, dim3((W * W + 255) / 256, wt.n), dim3(256), 0, s, sm, med, wt, X, Y, W, L, flags);
      TTC_HIP(c, hipGetLastError());
      const size_t lds = 3 * (size_t)W * W;
      static LdsConfig cfg_bright;
      TTC_HIP(c, cfg_bright.ensure(&k_bright_dist, lds));
      hipLaunchKernelGGL(k_bright_dist, dim3(wt.n), dim3(1024), lds, s, flags, W, size, d2);
      TTC_HIP(c, hipGetLastError()); }
    TTC_CHECK(model_forward_frames(c, wt.n, probs, s, blocked_frames ? FRAMES_B16 : FRAMES_PLANAR));
    { KTimer kt(c, "post", s);
      unsigned char* cc = static_cast<unsigned char*>(c->scratch_buf("tile_clear", (size_t)npix));
      if (!cc) return c->fail(TTC_ERR_NOMEM, "clear-count map");
      hipLaunchKernelGGL(k_clear_map, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, d_interp, d_wm, T, (int)npix, cc);
      PostArgs pa{probs, cc, d2, d_windows, d_windows_raw, wt, d_wm, T, X, Y, size, n_dates_ok};
      const size_t lds = 2 * (size_t)(size + 2) * (size + 2);
      hipLaunchKernelGGL(k_post, dim3(wt.n), dim3(1024), lds, s, pa);
      TTC_HIP(c, hipGetLastError()); }
    return TTC_OK;
}

ttc_status tile_process_subtiles(ttc_ctx* c, const float* d_s2, int T, int X, int Y, const float* h_wmat,
                                 const int32_t* h_keep, const float* d_interp, const float* d_s1, const float* d_dem,
                                 const double* h_min, const double* h_max, int size, int n_dates_ok, float* d_windows,
                                 float* d_windows_raw, hipStream_t s) {
    if (!d_s2 || !h_wmat || !d_interp || !d_s1 || !d_dem || !h_min || !h_max || !d_windows)
        return c->fail(TTC_ERR_ARG, "process_subtiles: null argument");
    if (T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "process_subtiles: T must be in [1, 32]");
    // the host-built operator travels through a small ring of pinned slots, so the copy is asynchronous and the slot of a
    // call still in flight is not overwritten by the next ones
    constexpr int kSlots = 8;
    WMat* h_ring = static_cast<WMat*>(c->pinned_buf("wmat_ring", sizeof(WMat) * kSlots));
    WMat* d_ring = static_cast<WMat*>(c->scratch_buf("wmat_ring", sizeof(WMat) * kSlots));
    if (!h_ring || !d_ring) return c->fail(TTC_ERR_NOMEM, "operator staging");
    const int slot = (c->wmat_slot++) % kSlots;
    // a slot is rewritten only after the copy that last read it has executed (a caller may enqueue more than kSlots calls
    // without synchronising)
    if (c->wmat_events.size() < (size_t)kSlots) c->wmat_events.resize(kSlots, nullptr);
    if (c->wmat_events[slot]) TTC_HIP(c, hipEventSynchronize(c->wmat_events[slot]));
    else TTC_HIP(c, hipEventCreateWithFlags(&c->wmat_events[slot], hipEventDisableTiming));
    WMat& wm = h_ring[slot];
    std::memset(&wm, 0, sizeof(WMat));
    wm.T = T; wm.keep = 0; wm.Tk = 0;
    for (int t = 0; t < T; ++t) if (!h_keep || h_keep[t]) { wm.keep |= 1u << t; wm.Tk++; }
    for (int k = 0; k < 12; ++k)
        for (int t = 0; t < T; ++t) wm.w[k * kMaxT + t] = ((wm.keep >> t) & 1u) ? h_wmat[k * T + t] : 0.0f;
    TTC_HIP(c, hipMemcpyAsync(d_ring + slot, &wm, sizeof(WMat), hipMemcpyHostToDevice, s));
    TTC_HIP(c, hipEventRecord(c->wmat_events[slot], s));
    return tile_core(c, d_s2, T, X, Y, d_ring + slot, d_interp, d_s1, d_dem, h_min, h_max, size, n_dates_ok, d_windows, d_windows_raw,
                     false, s);
}

// the same with the operator built on the device from the missing-pixel counts and the acquisition days: no host round trip.
// d_dates [T] int32 in device memory.  d_status (optional, device int32): [1] <- dates kept.
ttc_status tile_process_subtiles_dev(ttc_ctx* c, float* d_s2, int T, int X, int Y, const int32_t* d_dates, const float* d_interp,
                                     const float* d_s1, const float* d_dem, const double* h_min, const double* h_max, int size,
                                     float* d_windows, float* d_windows_raw, bool stop_after_inputs, hipStream_t s) {
    if (!d_s2 || !d_dates || !d_interp || !d_s1 || !d_dem || !h_min || !h_max) return c->fail(TTC_ERR_ARG, "process_subtiles: null argument");
    if (T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "process_subtiles: T must be in [1, 32]");
    int32_t* counts = static_cast<int32_t*>(c->scratch_buf("miss_counts", sizeof(int32_t) * kMaxT));
    WMat* d_wm = static_cast<WMat*>(c->scratch_buf("wmat_dev", sizeof(WMat)));
    double* d_minv = static_cast<double*>(c->scratch_buf("wmat_minv", sizeof(double) * 12 * 24));
    if (!counts || !d_wm || !d_minv) return c->fail(TTC_ERR_NOMEM, "operator scratch");
    if (!c->minv_ready) {
        double M[12][24];
        whittaker_monthly_matrix(M);
        TTC_HIP(c, hipMemcpy(d_minv, M, sizeof(M), hipMemcpyHostToDevice));
        c->minv_ready = true;
    }
    {   // interpolate_na_vals (job.py:1149) + id_missing_px(arr, 10) (job.py:1032), one pass
        TTC_HIP(c, hipMemsetAsync(counts, 0, sizeof(int32_t) * T, s));
        KTimer kt(c, "nanfix_counts", s);
        hipLaunchKernelGGL(k_nanfix_counts, dim3(128, T), dim3(256), 0, s, d_s2, X * Y, counts);
        TTC_HIP(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_build_wmat, dim3(1), dim3(64), 0, s, counts, (X * X) / 10 + ((X * X) % 10 ? 1 : 0), d_dates, T, d_minv, d_wm, c->spec_status);
    TTC_HIP(c, hipGetLastError());
    return tile_core(c, d_s2, T, X, Y, d_wm, d_interp, d_s1, d_dem, h_min, h_max, size, -1, d_windows, d_windows_raw, stop_after_inputs, s);
}
