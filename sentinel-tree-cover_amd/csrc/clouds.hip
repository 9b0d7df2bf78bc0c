// Multi-temporal cloud / shadow detection (the step immediately before the gap-fill):
//   identify_clouds_shadows   src/preprocessing/cloud_removal.py:1215-1677
//   detect_pfcp               src/preprocessing/cloud_removal.py:1109-1212
// Per-pixel temporal rules, small-radius binary morphology (4-connected = L1 ball, 8-connected = square,
// Euclidean threshold = disk), per-image order statistics (radix select) and per-image moments.  Every flag is a
// byte plane [T][X*Y]; the stack arrives in the reference layout [T, X, Y, 10] float32.
// The two ESA-WorldCover rasters the reference reads with rasterio (cloud_removal.py:735-771) are inputs (or NULL,
// which reproduces the reference's except-branches: no forest, no urban pixels).
#include "ttc_internal.h"
#include "radix_select.h"

using namespace ttcsel;

namespace {

constexpr int kMaxT = 32;
constexpr int kKMax = 1024;          // iterations of the adaptive "close" threshold a pixel can survive (0.0025 each)

struct DateWin {                     // host-built index sets of one date
    int sh_lo, sh_hi;                // shadow reference window [lo, hi)                       (CR.py:1268-1275)
    int ot_lo, ot_hi;                // "others" window of the cloud test [lo, hi)             (CR.py:1341-1349)
    int ncl, cl[3];                  // "close" dates                                          (CR.py:1350-1363)
    int wl[10], wh[10];              // the ten widened windows [wl, wh) \ {t}                 (CR.py:1384-1390)
};

#define IMG(t, p, c) img[((long)(t) * npix + (p)) * 10 + (c)]

__device__ __forceinline__ void isort(float* v, int n) {
    for (int i = 1; i < n; ++i) {
        const float x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
        v[j + 1] = x;
    }
}
__device__ __forceinline__ float median_sorted(const float* v, int n) {    // numpy: mean of the two middle values, float32
    if (n <= 0) return NAN;
    return (n & 1) ? v[n >> 1] : (v[(n >> 1) - 1] + v[n >> 1]) * 0.5f;
}

// ---------------------------------------------------------------------------------------------- morphology
// out = OR of in' over the L1 ball of radius r (binary_dilation with the cross, r iterations; border_value 0), in two
// passes instead of 2r^2 + 2r + 1 probes: (1) distance to the nearest set pixel along the row, capped at r + 1;
// (2) a pixel is inside the ball iff some row dy away has such a pixel within r - |dy|.
__global__ void k_rowdist(const unsigned char* __restrict__ in, int H, int W, int r, int invert, unsigned char* __restrict__ hd) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    in += (long)blockIdx.y * H * W; hd += (long)blockIdx.y * H * W;
    const int y = p / W, x = p % W;
    int best = r + 1;
    for (int d = 0; d <= r && best > r; ++d) {
        const int xl = x - d, xr = x + d;
        if ((xl >= 0 && ((in[y * W + xl] != 0) != (invert != 0))) || (xr < W && ((in[y * W + xr] != 0) != (invert != 0)))) best = d;
    }
    hd[p] = (unsigned char)best;
}
__global__ void k_coldist(const unsigned char* __restrict__ hd, int H, int W, int r, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    hd += (long)blockIdx.y * H * W; out += (long)blockIdx.y * H * W;
    const int y = p / W, x = p % W;
    bool any = false;
    for (int dy = -r; dy <= r; ++dy) {
        const int yy = y + dy;
        if (yy >= 0 && yy < H && (int)hd[yy * W + x] <= r - abs(dy)) { any = true; break; }
    }
    out[p] = any;
}
// (round 5) the two passes with FOUR outputs per thread along the scan axis: (2 r + 4) byte loads per four outputs instead of up to 2 r + 1
// each -- these kernels are bound by their load instructions.  Same sets (gapfill.hip: k_dil_rows4 / k_dil_cols4 are the single-plane twins).
constexpr int kDilR4 = 10;                // largest radius of the four-output forms
__global__ void k_rowdist4(const unsigned char* __restrict__ in, int H, int W, int r, int invert, unsigned char* __restrict__ hd) {
    const int nb = (W + 3) / 4;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= H * nb) return;
    in += (long)blockIdx.y * H * W; hd += (long)blockIdx.y * H * W;
    const int y = id / nb, x0 = 4 * (id % nb);
    unsigned hit = 0;                                       // bit j: pixel x0 - r + j is set
#pragma unroll
    for (int j = 0; j < 2 * kDilR4 + 4; ++j) {
        const int xx = x0 - r + j;
        if (j < 2 * r + 4 && xx >= 0 && xx < W && ((in[y * W + xx] != 0) != (invert != 0))) hit |= 1u << j;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (x0 + q >= W) break;
        int best = r + 1;
        for (int d = r; d >= 0; --d)
            if (((hit >> (r + q - d)) | (hit >> (r + q + d))) & 1u) best = d;
        hd[y * W + x0 + q] = (unsigned char)best;
    }
}
__global__ void k_coldist4(const unsigned char* __restrict__ hd, int H, int W, int r, unsigned char* __restrict__ out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ((H + 3) / 4) * W) return;
    hd += (long)blockIdx.y * H * W; out += (long)blockIdx.y * H * W;
    const int y0 = 4 * (id / W), x = id % W;
    int gv[2 * kDilR4 + 4];
#pragma unroll
    for (int j = 0; j < 2 * kDilR4 + 4; ++j) {
        const int yy = y0 - r + j;
        gv[j] = (j < 2 * r + 4 && yy >= 0 && yy < H) ? (int)hd[yy * W + x] : 255;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (y0 + q >= H) break;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 2 * kDilR4 + 1; ++j) {         // dy = j - r
            const int dy = j - r;
            if (j <= 2 * r) any |= gv[q + j] <= r - (dy < 0 ? -dy : dy);
        }
        out[(y0 + q) * W + x] = any;
    }
}
// the 3-D cross (binary_dilation with the 3-D cross, r iterations: up to 63 probes for r = 3 in the direct form) in three separable passes: k_rowdist4, then the L1 distance inside the plane
// (k_coldist_d4: min over dy of |dy| + row distance, r + 1 = none within r), then a date offset dt with |dt| + that distance <= r (k_tdist)
__global__ void k_coldist_d4(const unsigned char* __restrict__ hd, int H, int W, int r, unsigned char* __restrict__ d2) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ((H + 3) / 4) * W) return;
    hd += (long)blockIdx.y * H * W; d2 += (long)blockIdx.y * H * W;
    const int y0 = 4 * (id / W), x = id % W;
    int gv[2 * kDilR4 + 4];
#pragma unroll
    for (int j = 0; j < 2 * kDilR4 + 4; ++j) {
        const int yy = y0 - r + j;
        gv[j] = (j < 2 * r + 4 && yy >= 0 && yy < H) ? (int)hd[yy * W + x] : 255;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (y0 + q >= H) break;
        int best = r + 1;
#pragma unroll
        for (int j = 0; j < 2 * kDilR4 + 1; ++j) {
            const int dy = j - r;
            if (j <= 2 * r) best = min(best, gv[q + j] + (dy < 0 ? -dy : dy));
        }
        d2[(y0 + q) * W + x] = (unsigned char)best;
    }
}
__global__ void k_tdist(const unsigned char* __restrict__ d2, int T, int npix, int r, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int v[kMaxT];
#pragma unroll
    for (int t = 0; t < kMaxT; ++t) v[t] = t < T ? (int)d2[(long)t * npix + p] : 255;
#pragma unroll
    for (int t = 0; t < kMaxT; ++t) {
        bool any = false;
#pragma unroll
        for (int dt = -3; dt <= 3; ++dt) {
            const int tt = t + dt;                              // constant after unrolling: v[] stays in registers
            if (tt >= 0 && tt < kMaxT) any |= ((dt < 0 ? -dt : dt) <= r) && tt < T && (v[tt] + (dt < 0 ? -dt : dt) <= r);
        }
        if (t < T) out[(long)t * npix + p] = any;
    }
}
// column pass of the Euclidean-radius dilation (distance_transform_edt(1 - in) <= R): k_rowdist4 gives the distance along the row (R + 1 = none within R); a pixel is within the
// Euclidean radius iff some row dy away has dy^2 + distance^2 <= r2.  (The direct form probes (2 R + 1)^2 pixels.)
__global__ void k_coleuclid4(const unsigned char* __restrict__ hd, const int* __restrict__ counts, int H, int W, int R, int r2,
                             unsigned char* __restrict__ out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (id >= ((H + 3) / 4) * W) return;
    hd += (long)t * H * W; out += (long)t * H * W;
    const int y0 = 4 * (id / W), x = id % W;
    if (counts[t] == 0) {                                   // scipy's transform of a plane without background: measured from (-1, 0)
        for (int q = 0; q < 4 && y0 + q < H; ++q) out[(y0 + q) * W + x] = ((y0 + q + 1) * (y0 + q + 1) + x * x) <= r2;
        return;
    }
    int gv[2 * kDilR4 + 4];
#pragma unroll
    for (int j = 0; j < 2 * kDilR4 + 4; ++j) {
        const int yy = y0 - R + j;
        gv[j] = (j < 2 * R + 4 && yy >= 0 && yy < H) ? (int)hd[yy * W + x] : 255;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (y0 + q >= H) break;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 2 * kDilR4 + 1; ++j) {
            const int dy = j - R;
            if (j <= 2 * R) any |= gv[q + j] <= R && dy * dy + gv[q + j] * gv[q + j] <= r2;
        }
        out[(y0 + q) * W + x] = any;
    }
}
// 8-connected structure, r iterations = square of radius r
__global__ void k_dil_sq(const unsigned char* __restrict__ in, int H, int W, int r, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    in += (long)blockIdx.y * H * W; out += (long)blockIdx.y * H * W;
    const int y = p / W, x = p % W;
    bool any = false;
    for (int yy = max(0, y - r); yy <= min(H - 1, y + r) && !any; ++yy)
        for (int xx = max(0, x - r); xx <= min(W - 1, x + r); ++xx)
            if (in[yy * W + xx]) { any = true; break; }
    out[p] = any;
}
__global__ void k_plane_count(const unsigned char* __restrict__ in, int npix, int* __restrict__ counts) {
    const int t = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) c += in[(long)t * npix + p] != 0;
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[t], c);
}
// (distance_transform_edt(1 - in) <= R), r2 = R*R: k_rowdist4 + k_coleuclid4 above.  A plane without any set pixel has no background for
// scipy's transform, which then measures from the virtual point (-1, 0).
__global__ void k_u8_not(const unsigned char* __restrict__ in, long n, unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] == 0;
}
// 3x3 sum with numpy 'reflect' padding (CR.py:1247-1252)
__device__ __forceinline__ int winsum3(const unsigned char* pl, int H, int W, int y, int x) {
    int s = 0;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = y + dy; yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = x + dx; xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            s += pl[yy * W + xx] != 0;
        }
    }
    return s;
}

// ---------------------------------------------------------------------------------------------- stage 0/1: water, Hollstein mask
__global__ void k_water(const float* __restrict__ img, int T, int npix, float* __restrict__ water) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float v[kMaxT];
    int n = 0;
    for (int t = 0; t < T; ++t) {
        const float g = IMG(t, p, 1), nr = IMG(t, p, 3);
        const float w = (g - nr) / (g + nr);
        if (!isnan(w)) v[n++] = w;
    }
    isort(v, n);
    water[p] = median_sorted(v, n);
}
__global__ void k_hollstein(const float* __restrict__ img, int T, int npix, unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    out[(long)t * npix + p] = (IMG(t, p, 7) > 0.166f) && (IMG(t, p, 1) > 0.28f) && ((IMG(t, p, 5) / IMG(t, p, 8)) < 4.292f);
}

// ---------------------------------------------------------------------------------------------- stage 2: shadow candidates
// all-date references of bands {0, 1, 7, 8}: nanmedian over the dates the coarse mask leaves (fallback: plain median), and
// the plain minimum (fallback of the windowed median)
__global__ void k_allref(const float* __restrict__ img, const unsigned char* __restrict__ clm, int T, int npix,
                         float* __restrict__ all_med, float* __restrict__ all_min) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int bands[4] = {0, 1, 7, 8};
    for (int k = 0; k < 4; ++k) {
        float v[kMaxT], a[kMaxT];
        int n = 0;
        float mn = INFINITY;
        for (int t = 0; t < T; ++t) {
            const float x = IMG(t, p, bands[k]);
            a[t] = x; mn = fminf(mn, x);
            if (!clm[(long)t * npix + p]) v[n++] = x;
        }
        float m;
        if (n > 0) { isort(v, n); m = median_sorted(v, n); }
        else { isort(a, T); m = median_sorted(a, T); }
        all_med[(long)p * 4 + k] = m;
        all_min[(long)p * 4 + k] = mn;
    }
}
// ascending order of eight values in registers (Batcher's 19-comparator network): the window of k_shadow_cand holds at most eight dates, missing
// ones are +inf and end up behind the valid ones -- the same order the insertion sort of the first version produced in scratch memory
__device__ __forceinline__ void sort8(float (&v)[8]) {
#define TTC_CE(i, j) { const float lo_ = fminf(v[i], v[j]), hi_ = fmaxf(v[i], v[j]); v[i] = lo_; v[j] = hi_; }
    TTC_CE(0, 1) TTC_CE(2, 3) TTC_CE(4, 5) TTC_CE(6, 7)
    TTC_CE(0, 2) TTC_CE(1, 3) TTC_CE(4, 6) TTC_CE(5, 7)
    TTC_CE(1, 2) TTC_CE(5, 6)
    TTC_CE(0, 4) TTC_CE(1, 5) TTC_CE(2, 6) TTC_CE(3, 7)
    TTC_CE(2, 4) TTC_CE(3, 5)
    TTC_CE(1, 2) TTC_CE(3, 4) TTC_CE(5, 6)
#undef TTC_CE
}
__device__ __forceinline__ float pick8(const float (&v)[8], int i) {          // v[i] without dynamic register indexing
    float r = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = i == k ? v[k] : r;
    return r;
}
__global__ void k_shadow_cand(const float* __restrict__ img, const unsigned char* __restrict__ clm, const float* __restrict__ water,
                              const float* __restrict__ dem, const DateWin* __restrict__ wins, const float* __restrict__ all_med,
                              const float* __restrict__ all_min, int T, int npix, unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const DateWin& w = wins[t];
    const int bands[4] = {0, 1, 7, 8};
    float lmax[4], lmed[4];
    // (round 5) the window's values live in registers: eight predicated slots, a sorting network, selects -- the first version filled a
    // local array of runtime length and insertion-sorted it (scratch memory, data-dependent loops: 0.56 ms per tile)
    bool ok[8];
    int n = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tt = w.sh_lo + j;
        ok[j] = tt < w.sh_hi && !clm[(long)(tt < w.sh_hi ? tt : w.sh_lo) * npix + p];
        n += ok[j] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tt = w.sh_lo + j;
            v[j] = ok[j] ? IMG(tt, p, bands[k]) : INFINITY;
        }
        if (n > 0) {
            sort8(v);
            lmed[k] = (n & 1) ? pick8(v, n >> 1) : (pick8(v, (n >> 1) - 1) + pick8(v, n >> 1)) * 0.5f;
            lmax[k] = pick8(v, n - 1);
        } else { lmed[k] = all_min[(long)p * 4 + k]; lmax[k] = NAN; }
    }
    const float b = IMG(t, p, 0), g = IMG(t, p, 1), r = IMG(t, p, 2), a8 = IMG(t, p, 7), s11 = IMG(t, p, 8);
    const bool wet_px = water[p] > 0.f;
    const bool d8m = (a8 - lmax[2]) < -0.04f, d11m = (s11 - lmax[3]) < -0.04f;
    bool s = ((s11 - lmed[3]) < -0.04f) && ((a8 - lmed[2]) < -0.04f) && (b < 0.09f) && ((b - lmed[0]) < -0.02f) && (a8 < 0.17f);
    const bool dark = d11m && d8m && (b < 0.03f) && (a8 < 0.18f) && !wet_px;
    s = (s || dark) && !wet_px;
    const float sum3 = (b + g) + r;
    const bool slope = d8m && d11m && (b < 0.07f) && (a8 < 0.18f) && (sum3 < 0.28f) && !wet_px && (dem[p] >= 25.f);
    s = s || slope;
    const float* am = all_med + (long)p * 4;
    const bool wet = ((b - am[0]) < -0.05f) && ((g - am[1]) < -0.05f) && (a8 < 0.03f) && ((am[1] - g) > 0.02f) && wet_px;
    out[(long)t * npix + p] = s || wet;
}

// ---------------------------------------------------------------------------------------------- stage 4: cloud candidates
__global__ void k_extra_table(float* __restrict__ tab) {      // close_modifier += 0.0025 in float64, used as float32
    if (threadIdx.x || blockIdx.x) return;
    double e = 0.0;
    for (int k = 0; k < kKMax; ++k) { tab[k] = (float)e; e += 0.0025; }
}
__device__ __forceinline__ float pct25(float* v, int T) {     // np.percentile(., 25) of a float32 series (numpy's _lerp)
    isort(v, T);
    const double pos = 0.25 * (double)(T - 1);
    const int lo = (int)floor(pos);
    const double tt = pos - (double)lo;
    const float a = v[lo], b = v[lo + 1 < T ? lo + 1 : lo];
    const float d = b - a;
    const double r = tt < 0.5 ? (double)a + (double)d * tt : (double)b - (double)d * (1.0 - tt);
    return (float)r;
}
__global__ void k_cloud_refs(const float* __restrict__ img, const unsigned char* __restrict__ shadows, const unsigned char* __restrict__ forest,
                             const DateWin* __restrict__ wins, const float* __restrict__ extra, int T, int npix,
                             unsigned char* __restrict__ far, unsigned short* __restrict__ kstar, unsigned char* __restrict__ b75) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const DateWin& w = wins[t];
    float upper[3], nearv[3];
    if (T > 2) {
        bool got = false;
        upper[0] = upper[1] = upper[2] = INFINITY;
        for (int tt = w.ot_lo; tt < w.ot_hi; ++tt)
            if (!shadows[(long)tt * npix + p]) { got = true; for (int c = 0; c < 3; ++c) upper[c] = fminf(upper[c], IMG(tt, p, c)); }
        if (!got)
            for (int c = 0; c < 3; ++c) { float v[kMaxT]; for (int tt = 0; tt < T; ++tt) v[tt] = IMG(tt, p, c); upper[c] = pct25(v, T); }
        got = false;
        nearv[0] = nearv[1] = nearv[2] = INFINITY;
        for (int k = 0; k < w.ncl; ++k) {
            const int tt = w.cl[k];
            if (!shadows[(long)tt * npix + p]) { got = true; for (int c = 0; c < 3; ++c) nearv[c] = fminf(nearv[c], IMG(tt, p, c)); }
        }
        for (int it = 0; it < 10 && !got; ++it)
            for (int tt = w.wl[it]; tt < w.wh[it]; ++tt)
                if (tt != t && !shadows[(long)tt * npix + p]) { got = true; for (int c = 0; c < 3; ++c) nearv[c] = fminf(nearv[c], IMG(tt, p, c)); }
        if (!got)
            for (int c = 0; c < 3; ++c) { float m = INFINITY; for (int tt = 0; tt < T; ++tt) m = fminf(m, IMG(tt, p, c)); nearv[c] = m; }
    } else {
        for (int c = 0; c < 3; ++c) { float m = INFINITY; for (int tt = 0; tt < T; ++tt) m = fminf(m, IMG(tt, p, c)); nearv[c] = m; upper[c] = m; }
    }
    float thr = fminf(((nearv[0] / 0.02f) / 100.0f) + 0.005f, 0.10f);
    thr = fmaxf(thr, 0.05f);
    if (forest && forest[p] == 1) thr -= 0.02f;
    thr = fmaxf(thr, 0.04f);
    const float b = IMG(t, p, 0), g = IMG(t, p, 1), r = IMG(t, p, 2);
    far[(long)t * npix + p] = ((b - upper[0]) > 0.08f) && ((g - upper[1]) > 0.08f) && ((r - upper[2]) > 0.07f);
    const float d0 = b - nearv[0], d1 = g - nearv[1], d2 = r - nearv[2];
    int k = 0;
    for (; k < kKMax; ++k) {
        const float te = thr + extra[k];
        if (!((d0 > (te + 0.01f)) && (d1 > (te + 0.01f)) && (d2 > te))) break;
    }
    kstar[(long)t * npix + p] = (unsigned short)k;             // the pixel is a "close" cloud in iterations 0 .. k-1
    b75[(long)t * npix + p] = ((b + g) + r) < 0.75f;
}
// the adaptive loop of CR.py:1425-1441 replayed on per-iteration counts: histogram of the survival counts (many workgroups per
// date), then one thread per date walks the histogram
__global__ void k_kstar_hist(const unsigned char* __restrict__ far, const unsigned short* __restrict__ kstar, int npix,
                             int* __restrict__ hist /*[T][kKMax + 2]: bins, then the far count*/) {
    __shared__ int h[kKMax + 1];
    const int t = blockIdx.y;
    for (int i = threadIdx.x; i <= kKMax; i += blockDim.x) h[i] = 0;
    __syncthreads();
    int f = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        atomicAdd(&h[kstar[(long)t * npix + p]], 1);
        f += far[(long)t * npix + p] != 0;
    }
    for (int k = 32; k >= 1; k >>= 1) f += __shfl_xor(f, k);
    int* out = hist + (long)t * (kKMax + 2);
    if ((threadIdx.x & 63) == 0 && f) atomicAdd(&out[kKMax + 1], f);
    __syncthreads();
    for (int i = threadIdx.x; i <= kKMax; i += blockDim.x) if (h[i]) atomicAdd(&out[i], h[i]);
}
__global__ void k_cloud_loop(const int* __restrict__ hist_all, int T, int npix, int* __restrict__ kfinal) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int* hist = hist_all + (long)t * (kKMax + 2);
    const int nfar = hist[kKMax + 1];
    double fi = 0.0, fc = 1.0;
    int k = 0, remaining = npix - hist[0];                      // pixels with kstar > 0 are "close" clouds in iteration 0
    int used = 0;
    while ((fc - fi) > 0.075) {
        fi = (double)nfar / (double)npix;
        fc = (double)remaining / (double)npix;
        used = k;
        ++k;
        if (k > kKMax) break;
        remaining -= hist[k];                                    // pixels that survive iteration k have kstar > k
    }
    kfinal[t] = used;
}
__global__ void k_cloud_near(const unsigned short* __restrict__ kstar, const int* __restrict__ kfinal, const unsigned char* __restrict__ b75,
                             int npix, unsigned char* __restrict__ nearc) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const long i = (long)t * npix + p;
    nearc[i] = (kstar[i] > kfinal[t]) && b75[i];
}
// clouds = max(far, near), where outside forests `near` is first eroded by two 4-connected steps (CR.py:1446-1449)
__global__ void k_cloud_merge(const unsigned char* __restrict__ far, const unsigned char* __restrict__ nearc,
                              const unsigned char* __restrict__ near_bg_dil, const unsigned char* __restrict__ forest, int npix,
                              unsigned char* __restrict__ clouds) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const long i = (long)t * npix + p;
    const bool in_forest = forest && forest[p] != 0;
    const bool nc = in_forest ? (nearc[i] != 0) : (near_bg_dil[i] == 0);
    clouds[i] = far[i] || nc;
}

// ---------------------------------------------------------------------------------------------- stage 5: brightness outliers
// selection source: visible brightness of the pixels that are neither cloud nor shadow, one problem pair per date
struct SrcBright {
    const float* img; const unsigned char* clouds; const unsigned char* shadows; int npix;
    __device__ int count() const { return npix; }
    __device__ bool get(int q, int p, float& v) const {
#pragma clang fp contract(off)
        const int t = q >> 1;
        const long i = (long)t * npix + p;
        if (clouds[i] || shadows[i]) return false;
        const float* px = img + i * 10;
        v = (px[0] + px[1]) + px[2];
        return true;
    }
};
__global__ void k_bright_count(const unsigned char* __restrict__ clouds, const unsigned char* __restrict__ shadows, int npix,
                               int* __restrict__ n_free, int* __restrict__ n_clear) {
    const int t = blockIdx.y;
    int a = 0, b = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const long i = (long)t * npix + p;
        a += !(clouds[i] || shadows[i]); b += !clouds[i];
    }
    for (int k = 32; k >= 1; k >>= 1) { a += __shfl_xor(a, k); b += __shfl_xor(b, k); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&n_free[t], a); atomicAdd(&n_clear[t], b); }
}
__global__ void k_sel_init_median(SelState* __restrict__ st, const int* __restrict__ n, int T) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 2 * T) return;
    const int c = n[q >> 1];
    SelState ss; ss.prefix = 0; ss.mask = 0; ss.k = c > 0 ? ((q & 1) ? c / 2 : (c - 1) / 2) : 0;
    st[q] = ss;
}
// ratio = brightness / per-image median (1 over water); moments over the clear pixels (all pixels when none is clear)
__device__ __forceinline__ float bright_ratio(const float* img, const float* water, const SelState* st, const int* n_free, int npix, int t, int p) {
#pragma clang fp contract(off)
    const float* px = img + ((long)t * npix + p) * 10;
    const float med = n_free[t] > 0 ? (fkey_inv(st[2 * t].prefix) + fkey_inv(st[2 * t + 1].prefix)) * 0.5f : NAN;
    float r = ((px[0] + px[1]) + px[2]) / med;
    if (water[p] > 0.f) r = 1.0f;
    return r;
}
__global__ void k_bright_moments(const float* __restrict__ img, const float* __restrict__ water, const unsigned char* __restrict__ clouds,
                                 const SelState* __restrict__ st, const int* __restrict__ n_free, const int* __restrict__ n_clear,
                                 int npix, int pass, double* __restrict__ acc /*[T][3]: sum, count, sumsq*/) {
    const int t = blockIdx.y;
    const bool use_clear = n_clear[t] > 0;
    const double mean = pass ? acc[t * 3] / fmax(acc[t * 3 + 1], 1.0) : 0.0;
    double s = 0.0, c = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        if (use_clear && clouds[(long)t * npix + p]) continue;
        const float r = bright_ratio(img, water, st, n_free, npix, t, p);
        if (isnan(r)) continue;
        if (pass == 0) { s += (double)r; c += 1.0; }
        else { const double d = (double)r - mean; s += d * d; }
    }
    for (int k = 32; k >= 1; k >>= 1) { s += __shfl_xor(s, k); c += __shfl_xor(c, k); }
    if ((threadIdx.x & 63) == 0) {
        if (pass == 0) { atomicAdd(&acc[t * 3], s); atomicAdd(&acc[t * 3 + 1], c); }
        else atomicAdd(&acc[t * 3 + 2], s);
    }
}
__global__ void k_bright_flags(const float* __restrict__ img, const float* __restrict__ water, const SelState* __restrict__ st,
                               const int* __restrict__ n_free, const double* __restrict__ acc, int npix, unsigned char* __restrict__ bright) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const double cnt = acc[t * 3 + 1];
    const float mean = cnt > 0 ? (float)(acc[t * 3] / cnt) : NAN;
    const float sd = cnt > 0 ? (float)sqrt(acc[t * 3 + 2] / cnt) : NAN;
    const float r = bright_ratio(img, water, st, n_free, npix, t, p);
    const float z = (r - mean) / sd;
    bright[(long)t * npix + p] = (z > 3.5f) && (water[p] < 0.f);
}
__global__ void k_bright_merge(const unsigned char* __restrict__ bright, int T, int npix, unsigned char* __restrict__ clouds) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int rep = 0;
    for (int t = 0; t < T; ++t) rep += bright[(long)t * npix + p] && !clouds[(long)t * npix + p];
    if (rep > 1) return;
    for (int t = 0; t < T; ++t) if (bright[(long)t * npix + p]) clouds[(long)t * npix + p] = 1;
}
// stage 6: clouds are white
__global__ void k_white(const float* __restrict__ img, int npix, unsigned char* __restrict__ clouds) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const float b = IMG(t, p, 0), g = IMG(t, p, 1), r = IMG(t, p, 2);
    const float mean_b = ((b + g) + r) / 3.0f;
    const float rng = fmaxf(fmaxf(b, g), r) - fminf(fminf(b, g), r);
    if ((mean_b < 0.4f) && ((rng / mean_b) > 0.5f)) clouds[(long)t * npix + p] = 0;
}

// ---------------------------------------------------------------------------------------------- stage 7: detect_pfcp
// pfps = median_t(ndbi > 0 & ndbi > ndvi) * (median_t ndwi < 0), urban core -> 1, far from urban -> 0, high ground -> 0;
// flag = pfps != 0 (only its support matters after the dilation)
__global__ void k_pf_base(const float* __restrict__ img, const float* __restrict__ dem, const unsigned char* __restrict__ core,
                          const unsigned char* __restrict__ nearu, int T, int npix, unsigned char* __restrict__ pf) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float w[kMaxT];
    int nb = 0;
    bool nanw = false;
    for (int t = 0; t < T; ++t) {
        const float g = IMG(t, p, 1), r = IMG(t, p, 2), n = IMG(t, p, 3), s = IMG(t, p, 8);
        const float ndbi = (s - n) / (s + n), ndvi = (n - r) / (n + r);
        nb += (ndbi > 0.f) && (ndbi > ndvi);
        w[t] = (g - n) / (g + n);
        nanw |= isnan(w[t]);
    }
    // median of T booleans: > 0 when at least ceil(T / 2) ... of the sorted values the middle one(s) are not both 0
    const bool med_b = (T & 1) ? (2 * nb > T) : (2 * nb >= T);          // upper-middle element set => median in {0.5, 1}
    float mw = NAN;
    if (!nanw) { isort(w, T); mw = median_sorted(w, T); }
    bool v = med_b && (mw < 0.f);
    if (core[p] == 1) v = true;
    if (nearu[p] == 0) v = false;
    if ((dem[p] / 90.0f) > 0.10f) v = false;
    pf[p] = v;
}
// scipy.ndimage.gaussian_filter(sigma = 0.5, truncate = 3): 5 taps, 'reflect' (half-sample symmetric), float64 accumulate,
// float32 result after each axis; axis 0 then axis 1
__device__ __forceinline__ int refl(int i, int n) { while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i; return i; }
__global__ void k_gauss5(const float* __restrict__ src, long sstride /*elements between pixels*/, int H, int W, int axis,
                         float* __restrict__ dst) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    double wgt[5], ws = 0.0;
    for (int i = -2; i <= 2; ++i) { wgt[i + 2] = exp(-0.5 / (0.5 * 0.5) * (double)(i * i)); ws += wgt[i + 2]; }
    auto at = [&](int i) -> double {
        const int yy = axis == 0 ? refl(y + i, H) : y, xx = axis == 1 ? refl(x + i, W) : x;
        return (double)src[((long)t * H * W + (long)yy * W + xx) * sstride];
    };
    // NI_Correlate1D, symmetric filter: centre tap, then (x[-j] + x[+j]) * w_j from the outermost pair inwards
    double a = at(0) * (wgt[2] / ws);
    a += (at(-2) + at(2)) * (wgt[0] / ws);
    a += (at(-1) + at(1)) * (wgt[1] / ws);
    dst[(long)t * H * W + p] = (float)a;
}
// 2x2 block means of three planes -> ratios r8a = B8s / B8A, r8a7 = B7 / B8A at half resolution
__global__ void k_half_ratios(const float* __restrict__ img, const float* __restrict__ b8s, int H, int W, float* __restrict__ ra,
                              float* __restrict__ rb) {
#pragma clang fp contract(off)
    const int h2 = H / 2, w2 = W / 2, npix = H * W;
    const int q = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (q >= h2 * w2) return;
    const int y = 2 * (q / w2), x = 2 * (q % w2);
    const int p00 = y * W + x, p01 = p00 + 1, p10 = p00 + W, p11 = p10 + 1;
    const float* s = b8s + (long)t * npix;
    const float m8 = (((s[p00] + s[p01]) + s[p10]) + s[p11]) / 4.0f;
    const float m8a = (((IMG(t, p00, 7) + IMG(t, p01, 7)) + IMG(t, p10, 7)) + IMG(t, p11, 7)) / 4.0f;
    const float m7 = (((IMG(t, p00, 6) + IMG(t, p01, 6)) + IMG(t, p10, 6)) + IMG(t, p11, 6)) / 4.0f;
    ra[(long)t * h2 * w2 + q] = m8 / m8a;
    rb[(long)t * h2 * w2 + q] = m7 / m8a;
}
// local variance over 7x7 ('symm' boundary) of both ratios in float64, CDI test, nearest x2, NDVI < 0.4
__device__ __forceinline__ int symm(int i, int n) { while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - 1 - i; return i; }
__global__ void k_cdi(const float* __restrict__ ra, const float* __restrict__ rb, int h2, int w2, unsigned char* __restrict__ hit) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (q >= h2 * w2) return;
    const int y = q / w2, x = q % w2;
    const float* A = ra + (long)t * h2 * w2; const float* B = rb + (long)t * h2 * w2;
    double sa = 0, sa2 = 0, sb = 0, sb2 = 0;
    const double k = 1.0 / 49.0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const int i = symm(y + dy, h2) * w2 + symm(x + dx, w2);
            const float a = A[i], b = B[i];
            const float a2 = a * a, b2 = b * b;                       // r ** 2 is float32 before the convolution
            sa += (double)a * k; sa2 += (double)a2 * k; sb += (double)b * k; sb2 += (double)b2 * k;
        }
    const double va = sa2 - sa * sa, vb = sb2 - sb * sb;
    const double cdi = (vb - va) / (vb + va);
    hit[(long)t * h2 * w2 + q] = cdi >= -0.4;
}
__global__ void k_cdis(const float* __restrict__ img, const unsigned char* __restrict__ hit, int H, int W, unsigned char* __restrict__ cdis) {
#pragma clang fp contract(off)
    const int npix = H * W, p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const int y = p / W, x = p % W;
    const float n = IMG(t, p, 3), r = IMG(t, p, 2);
    cdis[(long)t * npix + p] = hit[(long)t * (H / 2) * (W / 2) + (y / 2) * (W / 2) + x / 2] && (((n - r) / (n + r)) < 0.4f);
}
__global__ void k_and_planes(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, long n, unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] && b[i];
}
__global__ void k_tile_plane(const unsigned char* __restrict__ a, int T, int npix, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    for (int t = 0; t < T; ++t) out[(long)t * npix + p] = a[p];
}

// ---------------------------------------------------------------------------------------------- stage 8: false-positive rules
__device__ __forceinline__ bool not_much_brighter(const float* img, int T, int npix, int t, int p) {
#pragma clang fp contract(off)
    float floor_v = INFINITY;
    for (int tt = max(t - 1, 0); tt < min(t + 2, T); ++tt)
        for (int c = 0; c < 3; ++c) floor_v = fminf(floor_v, IMG(tt, p, c));
    const float mean_b = ((IMG(t, p, 0) + IMG(t, p, 1)) + IMG(t, p, 2)) / 3.0f;
    return (mean_b - floor_v) < 0.4f;
}
__global__ void k_fp_urban(const float* __restrict__ img, const unsigned char* __restrict__ fcps, int T, int npix,
                           unsigned char* __restrict__ clouds, unsigned char* __restrict__ shadows) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const long i = (long)t * npix + p;
    if (fcps[i] && not_much_brighter(img, T, npix, t, p)) { clouds[i] = 0; shadows[i] = 0; }
}
__global__ void k_nsr(const float* __restrict__ img, int npix, unsigned char* __restrict__ nsr) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    nsr[(long)t * npix + p] = (IMG(t, p, 3) / (IMG(t, p, 8) + 0.01f)) < 0.75f;
}
__global__ void k_fp_nsr(const float* __restrict__ img, const float* __restrict__ water, int T, int npix, unsigned char* __restrict__ nsr,
                         unsigned char* __restrict__ clouds) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const long i = (long)t * npix + p;
    if (water[p] < 0.f) nsr[i] = 0;
    if (nsr[i] && not_much_brighter(img, T, npix, t, p)) clouds[i] = 0;
}
__global__ void k_water_dark(const float* __restrict__ img, const float* __restrict__ water, int npix, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    out[(long)t * npix + p] = (water[p] > 0.f) && (IMG(t, p, 8) < 0.11f);
}
__global__ void k_clear_where(const unsigned char* __restrict__ mask, long n, unsigned char* __restrict__ clouds) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && mask[i]) clouds[i] = 0;
}
__global__ void k_lone(const unsigned char* __restrict__ in, int H, int W, int least, unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= H * W) return;
    const unsigned char* pl = in + (long)t * H * W;
    out[(long)t * H * W + p] = pl[p] && winsum3(pl, H, W, p / W, p % W) >= least;
}
__global__ void k_dark_px(const float* __restrict__ img, int npix, unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    out[(long)t * npix + p] = ((IMG(t, p, 0) + IMG(t, p, 1)) + IMG(t, p, 2)) < 0.21f;
}
// CR.py:1563-1567 indexes clouds[i] with a uint8 ARRAY (fancy indexing, not a mask): the values present in
// (dilated dark pixels) * (1 - forest) select whole ROWS -- row 0 when any value is 0, row 1 when any value is 1
__global__ void k_dark_rows_flags(const unsigned char* __restrict__ dil, const unsigned char* __restrict__ forest, int npix, int* __restrict__ has01) {
    const int t = blockIdx.y;
    int a = 0, b = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int v = dil[(long)t * npix + p] && !(forest && forest[p]);
        a |= v == 0; b |= v == 1;
    }
    if (__any(a)) { if ((threadIdx.x & 63) == 0) atomicOr(&has01[2 * t], 1); }
    if (__any(b)) { if ((threadIdx.x & 63) == 0) atomicOr(&has01[2 * t + 1], 1); }
}
__global__ void k_dark_rows_apply(const int* __restrict__ has01, int H, int W, unsigned char* __restrict__ clouds) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (x >= W) return;
    if (has01[2 * t]) clouds[(long)t * H * W + x] = 0;
    if (has01[2 * t + 1] && H > 1) clouds[(long)t * H * W + W + x] = 0;
}

// ---------------------------------------------------------------------------------------------- stage 9: shape clean-up
__global__ void k_split_urban(const unsigned char* __restrict__ clouds, const unsigned char* __restrict__ pf, long n,
                              unsigned char* __restrict__ urban, unsigned char* __restrict__ rest) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    urban[i] = clouds[i] && pf[i];
    rest[i] = clouds[i] && !pf[i];
}
__global__ void k_big_small(const unsigned char* __restrict__ rest, int H, int W, unsigned char* __restrict__ big, unsigned char* __restrict__ small) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= H * W) return;
    const unsigned char* pl = rest + (long)t * H * W;
    const int n9 = winsum3(pl, H, W, p / W, p % W);
    big[(long)t * H * W + p] = pl[p] && n9 >= 6;
    small[(long)t * H * W + p] = pl[p] && n9 < 6;
}
__global__ void k_or_planes(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, long n, unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] || b[i];
}
// ---------------------------------------------------------------------------------------------- stage 10: implausible shadow amounts
__global__ void k_mean_flags(const unsigned char* __restrict__ a, int npix, int* __restrict__ counts) {   // alias of k_plane_count (separate buffer)
    const int t = blockIdx.y;
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) c += a[(long)t * npix + p] != 0;
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[t], c);
}
// keep shadows only within 50 4-connected steps of a cloud (or on ground >= 30 m) for the dates the two rules select
__global__ void k_shadow_limit(const unsigned char* __restrict__ clouds, const float* __restrict__ dem, const int* __restrict__ n_sh,
                               const int* __restrict__ n_cl, int H, int W, unsigned char* __restrict__ shadows) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    const int npix = H * W;
    if (p >= npix) return;
    const double ms = (double)n_sh[t] / npix, mc = (double)n_cl[t] / npix;
    // float32 means in the reference (np.mean of a float32 plane); the comparisons below are far from ties for 0/1 planes
    const bool rule1 = ((float)ms > (float)mc + 0.3f) && ((float)mc < 0.3f);
    // rule 2 is evaluated after rule 1 may have removed shadows; both rules apply the same mask, so applying it once is the same
    const bool rule2 = ((float)mc < 0.05f) && (((float)ms / (float)mc) > 3.0f);
    if (!(rule1 || rule2)) return;
    const long i = (long)t * npix + p;
    if (!shadows[i] || dem[p] >= 30.f) return;
    const unsigned char* pl = clouds + (long)t * npix;
    const int y = p / W, x = p % W, r = 50;
    for (int dy = -r; dy <= r; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const int rx = r - abs(dy);
        for (int dx = max(-rx, -x); dx <= min(rx, W - 1 - x); ++dx)
            if (pl[yy * W + x + dx]) return;
    }
    shadows[i] = 0;
}
// ---------------------------------------------------------------------------------------------- stage 11: extra shadows from blue statistics
__global__ void k_inv_blue_moments(const float* __restrict__ img, const unsigned char* __restrict__ clouds, int npix, int pass,
                                   double* __restrict__ acc /*[T][3]*/) {
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    const double mean = pass ? acc[t * 3] / fmax(acc[t * 3 + 1], 1.0) : 0.0;
    double s = 0.0, c = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        if (clouds[(long)t * npix + p]) continue;
        const float inv = 1.0f / IMG(t, p, 0);
        if (pass == 0) { s += (double)inv; c += 1.0; } else { const double d = (double)inv - mean; s += d * d; }
    }
    for (int k = 32; k >= 1; k >>= 1) { s += __shfl_xor(s, k); c += __shfl_xor(c, k); }
    if ((threadIdx.x & 63) == 0) { if (pass == 0) { atomicAdd(&acc[t * 3], s); atomicAdd(&acc[t * 3 + 1], c); } else atomicAdd(&acc[t * 3 + 2], s); }
}
__global__ void k_extra_shadow(const float* __restrict__ img, const double* __restrict__ acc, const int* __restrict__ n_cl, int npix,
                               unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    const long i = (long)t * npix + p;
    const bool active = ((float)((double)n_cl[t] / npix)) < 0.9f;
    const double cnt = acc[t * 3 + 1];
    const float level = (float)(acc[t * 3] / cnt) + 2.0f * (float)sqrt(acc[t * 3 + 2] / cnt);
    out[i] = active && ((1.0f / IMG(t, p, 0)) > level) && (IMG(t, p, 7) < 0.17f);
}
__global__ void k_extra_merge(const unsigned char* __restrict__ extra, const float* __restrict__ water, const int* __restrict__ n_cl, int npix,
                              unsigned char* __restrict__ clouds) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p >= npix) return;
    if (!(((float)((double)n_cl[t] / npix)) < 0.9f)) return;
    const long i = (long)t * npix + p;
    if (extra[i] && !(water[p] > 0.f)) clouds[i] = 1;
}
// ---------------------------------------------------------------------------------------------- stage 12: haze
__global__ void k_haze_moments(const float* __restrict__ img, const unsigned char* __restrict__ clouds, int npix, int pass,
                               double* __restrict__ acc /*[T][6]: sum_b, n, sum_w, ssq_b, ssq_w, -*/) {
#pragma clang fp contract(off)
    const int t = blockIdx.y;
    const double n0 = fmax(acc[t * 6 + 1], 1.0);
    const double mb = pass ? acc[t * 6] / n0 : 0.0, mw = pass ? acc[t * 6 + 2] / n0 : 0.0;
    double a = 0, b = 0, c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        if (clouds[(long)t * npix + p]) continue;
        const float bl = IMG(t, p, 0), g = IMG(t, p, 1), r = IMG(t, p, 2);
        const float mean_b = ((bl + g) + r) / 3.0f;
        const float ptp = fmaxf(fmaxf(bl, g), r) - fminf(fminf(bl, g), r);
        if (pass == 0) { a += (double)mean_b; b += 1.0; c += (double)ptp; }
        else { const double d1 = (double)mean_b - mb, d2 = (double)ptp - mw; a += d1 * d1; c += d2 * d2; }
    }
    for (int k = 32; k >= 1; k >>= 1) { a += __shfl_xor(a, k); b += __shfl_xor(b, k); c += __shfl_xor(c, k); }
    if ((threadIdx.x & 63) == 0) {
        if (pass == 0) { atomicAdd(&acc[t * 6], a); atomicAdd(&acc[t * 6 + 1], b); atomicAdd(&acc[t * 6 + 2], c); }
        else { atomicAdd(&acc[t * 6 + 3], a); atomicAdd(&acc[t * 6 + 4], c); }
    }
}
__device__ __forceinline__ float median_small(float* v, int n) { isort(v, n); return median_sorted(v, n); }
// per-image brightness / flatness / whiteness against the medians over the images that still have clear pixels; note the
// reference indexes the images to blank by their position in THAT list (CR.py:1672-1676)
__global__ void k_haze_decide(const double* __restrict__ acc, int T, int npix, int* __restrict__ hazy) {
#pragma clang fp contract(off)
    if (threadIdx.x || blockIdx.x) return;
    float mb[kMaxT], sb[kMaxT], sw[kMaxT], tmp[kMaxT];
    int n = 0;
    for (int t = 0; t < T; ++t) {
        hazy[t] = 0;
        const double c = acc[t * 6 + 1];
        if (!(c > 0.0)) continue;                               // np.mean(clouds[i]) < 1
        mb[n] = (float)(acc[t * 6] / c); sb[n] = (float)sqrt(acc[t * 6 + 3] / c); sw[n] = (float)sqrt(acc[t * 6 + 4] / c);
        ++n;
    }
    if (n == 0) return;
    for (int i = 0; i < n; ++i) tmp[i] = mb[i];
    const float med_b = median_small(tmp, n);
    for (int i = 0; i < n; ++i) tmp[i] = sb[i];
    const float med_s = median_small(tmp, n);
    for (int i = 0; i < n; ++i) tmp[i] = sw[i];
    const float med_w = median_small(tmp, n);
    for (int i = 0; i < n; ++i) {
        const float hb = mb[i] / med_b, hs = sb[i] / med_s, hw = sw[i] / med_w;
        hazy[i] = ((hb >= 1.5f) && (hs <= 0.67f) && (hw < 1.0f)) || ((hb >= 1.3f) && (hs <= 0.5f));
    }
}
__global__ void k_haze_apply(const int* __restrict__ hazy, int npix, unsigned char* __restrict__ clouds) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (p < npix && hazy[t]) clouds[(long)t * npix + p] = 1;
}
__global__ void k_u8_to_f32(const unsigned char* __restrict__ in, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ? 1.0f : 0.0f;
}

}  // namespace

// -------------------------------------------------------------------------------------------------------------------
static void build_windows(int T, std::vector<DateWin>& out) {
    out.assign(T, DateWin{});
    for (int t = 0; t < T; ++t) {
        DateWin& w = out[t];
        int lo = std::max(0, t - 4), hi = std::min(T, t + 3);
        if (hi - lo == 3) { if (hi == T) lo = std::max(lo - 1, 0); if (lo == 0) hi = std::min(hi + 1, T); }
        w.sh_lo = lo; w.sh_hi = hi;
        lo = std::max(0, t - 2); hi = std::min(T, t + 3);
        if (hi - lo == 3) { if (hi == T) lo = std::max(lo - 2, 0); if (lo == 0) hi = std::min(hi + 2, T); }
        w.ot_lo = lo; w.ot_hi = hi;
        int c0 = std::max(0, t - 1), c1 = std::min(T - 1, t + 1);
        if (c1 - c0 < 2) { if (c0 == 0) { ++c0; ++c1; } else { --c0; --c1; } }
        w.ncl = 0;
        if (c1 >= T - 2 && T > 3) w.cl[w.ncl++] = c0 - 1;
        w.cl[w.ncl++] = c0; w.cl[w.ncl++] = c1;
        int mn = w.cl[0], mx = w.cl[w.ncl - 1];
        for (int it = 0; it < 10; ++it) { mn = std::max(mn - 1, 0); mx = std::min(mx + 1, T); w.wl[it] = mn; w.wh[it] = mx; }
        for (int k = 0; k < w.ncl; ++k) w.cl[k] = std::min(std::max(w.cl[k], 0), T - 1);
    }
}

ttc_status clouds_identify(ttc_ctx* c, const float* img, int T, int X, int Y, const float* dem, const uint8_t* forest,
                           const uint8_t* urban_core, const uint8_t* urban_near, float* d_clouds, uint8_t* d_fcps, int debug_stage,
                           hipStream_t s) {
    if (!img || !dem || !d_clouds || !d_fcps || T < 1 || T > kMaxT || X < 2 || Y < 2)
        return c->fail(TTC_ERR_ARG, "identify_clouds_shadows: bad argument (T in [1, 32])");
    if ((urban_core == nullptr) != (urban_near == nullptr)) return c->fail(TTC_ERR_ARG, "identify_clouds_shadows: give both urban masks or none");
    const bool urban = urban_core != nullptr;
    if (urban && ((X | Y) & 1)) return c->fail(TTC_ERR_ARG, "identify_clouds_shadows: the parallax test needs even tile sides");
    const int H = X, W = Y, npix = X * Y;
    const long N = (long)T * npix;
    constexpr int NPL = 12;
    unsigned char* arena = static_cast<unsigned char*>(c->scratch_buf("cd_planes", (size_t)NPL * N));
    float* fbuf = static_cast<float*>(c->scratch_buf("cd_float", sizeof(float) * ((size_t)npix * 9 + (size_t)N * 3 + kKMax)));
    unsigned short* kstar = static_cast<unsigned short*>(c->scratch_buf("cd_kstar", sizeof(unsigned short) * (size_t)N));
    char* ctl = static_cast<char*>(c->scratch_buf("cd_ctl", 65536));
    if (!arena || !fbuf || !kstar || !ctl) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    unsigned char* P[NPL];
    for (int i = 0; i < NPL; ++i) P[i] = arena + (size_t)i * N;
    unsigned char *clm = P[0], *shadows = P[1], *clouds = P[2], *t1 = P[3], *t2 = P[4], *t3 = P[5], *far = P[6], *b75 = P[7],
                  *nsr = P[8], *fcps = P[9], *pfps = P[10], *twos = P[11];
    float* water = fbuf;                        // [npix]
    float* all_med = water + npix;              // [npix][4]
    float* all_min = all_med + 4L * npix;       // [npix][4]
    float* extra = all_min + 4L * npix;         // [kKMax]
    float* fa = extra + kKMax;                  // [T][npix] floats x3 (gaussian passes, half-res ratios)
    float* fb = fa + N; float* fc = fb + N;
    DateWin* d_wins = static_cast<DateWin*>(c->scratch_buf("cd_wins", sizeof(DateWin) * kMaxT));   // its own buffer: survives between calls
    int* cnt_a = reinterpret_cast<int*>(ctl + 8192);                         // [kMaxT] scratch counters
    int* cnt_b = cnt_a + kMaxT; int* cnt_c = cnt_b + kMaxT; int* kfinal = cnt_c + kMaxT; int* has01 = kfinal + kMaxT;   // has01 [2*kMaxT]
    int* hazy = has01 + 2 * kMaxT;
    SelState* st = reinterpret_cast<SelState*>(ctl + 16384);                 // 2 * kMaxT problems
    double* acc = reinterpret_cast<double*>(ctl + 20480);                    // [kMaxT][6]
    unsigned* hist = reinterpret_cast<unsigned*>(ctl + 24576);               // 64 * 256 * 4 = 65536 ... separate buffer below
    unsigned* hist_big = static_cast<unsigned*>(c->scratch_buf("cd_hist", sizeof(unsigned) * 2 * kMaxT * 256));
    if (!hist_big) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    (void)hist;
    if (!d_wins) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    static const int every_call = [] { const char* e = getenv("TTC_CD_WINS_EVERY_CALL"); return e ? atoi(e) : 0; }();   // probe: the old behaviour
    if (c->cd_wins_T != T || every_call) {
        // The table depends on T only: uploaded when T changes, NOT per call -- the stream synchronisation that makes the host vector
        // reusable waited for everything queued on this stream, i.e. for the previous tile of a pipelined tile loop (measured:
        // 8.4 ms of host time per tile inside the enqueue of job.predict_tiles, the GPU idle 20 % of the loop)
        std::vector<DateWin> h_wins;
        build_windows(T, h_wins);
        TTC_HIP(c, hipMemcpyAsync(d_wins, h_wins.data(), sizeof(DateWin) * T, hipMemcpyHostToDevice, s));
        TTC_HIP(c, hipStreamSynchronize(s));
        c->cd_wins_T = T;
    }

    KTimer kt(c, "identify_clouds", s);
    const dim3 b256(256), gp((npix + 255) / 256), gpt((npix + 255) / 256, T), gn((unsigned)((N + 255) / 256)), gred(32, T);
    auto zero_counts = [&](int* p) { return hipMemsetAsync(p, 0, sizeof(int) * kMaxT, s); };
    auto finish = [&](const unsigned char* plane) -> ttc_status {       // debug exit: return `plane` as the cloud output
        hipLaunchKernelGGL(k_u8_to_f32, gn, b256, 0, s, plane, N, d_clouds);
        TTC_HIP(c, hipMemsetAsync(d_fcps, 0, N, s));
        TTC_HIP(c, hipGetLastError());
        return TTC_OK;
    };
    unsigned char* hd = static_cast<unsigned char*>(c->scratch_buf("cd_rowdist", (size_t)N));
    if (!hd) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    auto dil_l1 = [&](const unsigned char* in, int r, int invert, unsigned char* out) {        // out may alias in
        if (r <= kDilR4) {
            hipLaunchKernelGGL(k_rowdist4, dim3((unsigned)(((long)H * ((W + 3) / 4) + 255) / 256), T), b256, 0, s, in, H, W, r, invert, hd);
            hipLaunchKernelGGL(k_coldist4, dim3((unsigned)(((long)((H + 3) / 4) * W + 255) / 256), T), b256, 0, s, hd, H, W, r, out);
        } else {
            hipLaunchKernelGGL(k_rowdist, gpt, b256, 0, s, in, H, W, r, invert, hd);
            hipLaunchKernelGGL(k_coldist, gpt, b256, 0, s, hd, H, W, r, out);
        }
    };
    unsigned char* hd2 = static_cast<unsigned char*>(c->scratch_buf("cd_coldist", (size_t)N));
    if (!hd2) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    auto dil_l1_3d = [&](const unsigned char* in, int r, unsigned char* out) -> ttc_status {   // r <= 3 (k_tdist's date window); in and out distinct
        if (r > 3) return c->fail(TTC_ERR_ARG, "cloud detection: the 3-D cross dilation is built for r <= 3");
        hipLaunchKernelGGL(k_rowdist4, dim3((unsigned)(((long)H * ((W + 3) / 4) + 255) / 256), T), b256, 0, s, in, H, W, r, 0, hd);
        hipLaunchKernelGGL(k_coldist_d4, dim3((unsigned)(((long)((H + 3) / 4) * W + 255) / 256), T), b256, 0, s, hd, H, W, r, hd2);
        hipLaunchKernelGGL(k_tdist, gp, b256, 0, s, hd2, T, npix, r, out);
        return TTC_OK;
    };
    auto near_euclid = [&](const unsigned char* in, const int* counts, int R, int r2, unsigned char* out) {     // R <= kDilR4; out may alias in
        hipLaunchKernelGGL(k_rowdist4, dim3((unsigned)(((long)H * ((W + 3) / 4) + 255) / 256), T), b256, 0, s, in, H, W, R, 0, hd);
        hipLaunchKernelGGL(k_coleuclid4, dim3((unsigned)(((long)((H + 3) / 4) * W + 255) / 256), T), b256, 0, s, hd, counts, H, W, R, r2, out);
    };
    // opening idiom of the reference: dilate(1 - dilate(x == 0, a), b)
    auto open_planes = [&](const unsigned char* in, int a, int b, unsigned char* tmp, unsigned char* out) {
        dil_l1(in, a, 1, tmp);
        dil_l1(tmp, b, 1, out);
    };

    // ---- 0/1: water index, coarse single-date mask
    hipLaunchKernelGGL(k_water, gp, b256, 0, s, img, T, npix, water);
    hipLaunchKernelGGL(k_hollstein, gpt, b256, 0, s, img, T, npix, t1);
    open_planes(t1, 2, 10, t2, clm);
    if (debug_stage == 1) return finish(clm);
    // ---- 2/3: shadows
    hipLaunchKernelGGL(k_allref, gp, b256, 0, s, img, clm, T, npix, all_med, all_min);
    hipLaunchKernelGGL(k_shadow_cand, gpt, b256, 0, s, img, clm, water, dem, d_wins, all_med, all_min, T, npix, t1);
    if (debug_stage == 2) return finish(t1);
    open_planes(t1, 2, 3, t2, t3);
    TTC_HIP(c, zero_counts(cnt_a));
    hipLaunchKernelGGL(k_plane_count, gred, b256, 0, s, t3, npix, cnt_a);
    near_euclid(t3, cnt_a, 5, 25, shadows);
    if (debug_stage == 3) return finish(shadows);
    // ---- 4: cloud candidates
    hipLaunchKernelGGL(k_extra_table, dim3(1), dim3(64), 0, s, extra);
    hipLaunchKernelGGL(k_cloud_refs, gpt, b256, 0, s, img, shadows, forest, d_wins, extra, T, npix, far, kstar, b75);
    int* khist = static_cast<int*>(c->scratch_buf("cd_khist", sizeof(int) * (size_t)kMaxT * (kKMax + 2)));
    if (!khist) return c->fail(TTC_ERR_NOMEM, "cloud detection scratch");
    TTC_HIP(c, hipMemsetAsync(khist, 0, sizeof(int) * (size_t)kMaxT * (kKMax + 2), s));
    hipLaunchKernelGGL(k_kstar_hist, gred, b256, 0, s, far, kstar, npix, khist);
    hipLaunchKernelGGL(k_cloud_loop, dim3(1), dim3(64), 0, s, khist, T, npix, kfinal);
    hipLaunchKernelGGL(k_cloud_near, gpt, b256, 0, s, kstar, kfinal, b75, npix, t1);
    dil_l1(t1, 2, 1, t2);
    hipLaunchKernelGGL(k_cloud_merge, gpt, b256, 0, s, far, t1, t2, forest, npix, clouds);
    if (debug_stage == 4) return finish(clouds);
    // ---- 5: brightness outliers
    TTC_HIP(c, zero_counts(cnt_a)); TTC_HIP(c, zero_counts(cnt_b));
    hipLaunchKernelGGL(k_bright_count, gred, b256, 0, s, clouds, shadows, npix, cnt_a, cnt_b);
    hipLaunchKernelGGL(k_sel_init_median, dim3(1), dim3(64), 0, s, st, cnt_a, T);
    TTC_HIP(c, hipMemsetAsync(hist_big, 0, sizeof(unsigned) * 2 * kMaxT * 256, s));
    TTC_HIP(c, radix_select(SrcBright{img, clouds, shadows, npix}, st, hist_big, 2 * T, s));
    TTC_HIP(c, hipMemsetAsync(acc, 0, sizeof(double) * kMaxT * 6, s));
    hipLaunchKernelGGL(k_bright_moments, gred, b256, 0, s, img, water, clouds, st, cnt_a, cnt_b, npix, 0, acc);
    hipLaunchKernelGGL(k_bright_moments, gred, b256, 0, s, img, water, clouds, st, cnt_a, cnt_b, npix, 1, acc);
    hipLaunchKernelGGL(k_bright_flags, gpt, b256, 0, s, img, water, st, cnt_a, acc, npix, t1);
    hipLaunchKernelGGL(k_bright_merge, gp, b256, 0, s, t1, T, npix, clouds);
    if (debug_stage == 5) return finish(clouds);
    // ---- 6: whiteness
    hipLaunchKernelGGL(k_white, gpt, b256, 0, s, img, npix, clouds);
    if (debug_stage == 6) return finish(clouds);
    // ---- 7: urban false-positive candidates (Fmask 4.0 parallax + built-up index)
    if (urban) {
        const int h2 = H / 2, w2 = W / 2;
        const dim3 gh((h2 * w2 + 255) / 256, T);
        hipLaunchKernelGGL(k_pf_base, gp, b256, 0, s, img, dem, urban_core, urban_near, T, npix, t1);
        hipLaunchKernelGGL(k_tile_plane, gp, b256, 0, s, t1, T, npix, t2);
        hipLaunchKernelGGL(k_dil_sq, gpt, b256, 0, s, t2, H, W, 6, pfps);
        hipLaunchKernelGGL(k_gauss5, gpt, b256, 0, s, img + 3, 10L, H, W, 0, fa);
        hipLaunchKernelGGL(k_gauss5, gpt, b256, 0, s, fa, 1L, H, W, 1, fb);
        hipLaunchKernelGGL(k_half_ratios, gh, b256, 0, s, img, fb, H, W, fa, fc);
        hipLaunchKernelGGL(k_cdi, gh, b256, 0, s, fa, fc, h2, w2, t1);
        hipLaunchKernelGGL(k_cdis, gpt, b256, 0, s, img, t1, H, W, t2);
        hipLaunchKernelGGL(k_dil_sq, gpt, b256, 0, s, t2, H, W, 6, t3);
        hipLaunchKernelGGL(k_and_planes, gn, b256, 0, s, pfps, t3, N, fcps);
    } else {
        TTC_HIP(c, hipMemsetAsync(pfps, 0, N, s));
        TTC_HIP(c, hipMemsetAsync(fcps, 0, N, s));
    }
    if (debug_stage == 7) return finish(fcps);
    if (debug_stage == 70) return finish(pfps);
    // ---- 8: false-positive rules
    hipLaunchKernelGGL(k_fp_urban, gpt, b256, 0, s, img, fcps, T, npix, clouds, shadows);
    hipLaunchKernelGGL(k_nsr, gpt, b256, 0, s, img, npix, t1);
    TTC_CHECK(dil_l1_3d(t1, 3, nsr));
    hipLaunchKernelGGL(k_fp_nsr, gpt, b256, 0, s, img, water, T, npix, nsr, clouds);
    hipLaunchKernelGGL(k_water_dark, gpt, b256, 0, s, img, water, npix, t1);
    dil_l1(t1, 10, 0, t2);
    hipLaunchKernelGGL(k_clear_where, gn, b256, 0, s, t2, N, clouds);
    hipLaunchKernelGGL(k_lone, gpt, b256, 0, s, clouds, H, W, 5, t1);
    hipLaunchKernelGGL(k_dark_px, gpt, b256, 0, s, img, npix, t2);
    dil_l1(t2, 3, 0, t3);
    TTC_HIP(c, hipMemsetAsync(has01, 0, sizeof(int) * 2 * kMaxT, s));
    hipLaunchKernelGGL(k_dark_rows_flags, gred, b256, 0, s, t3, forest, npix, has01);
    hipLaunchKernelGGL(k_dark_rows_apply, dim3((W + 255) / 256, T), b256, 0, s, has01, H, W, t1);
    TTC_HIP(c, hipMemcpyAsync(clouds, t1, N, hipMemcpyDeviceToDevice, s));
    if (debug_stage == 8) return finish(clouds);
    if (debug_stage == 80) return finish(shadows);
    if (debug_stage == 81) return finish(nsr);
    // ---- 9: shape clean-up
    dil_l1(clouds, 1, 1, t1);          // dilate(clouds == 0, 1)
    hipLaunchKernelGGL(k_u8_not, gn, b256, 0, s, t1, N, clouds);                     // eroded clouds
    dil_l1(pfps, 5, 0, t1);
    TTC_HIP(c, hipMemcpyAsync(pfps, t1, N, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_split_urban, gn, b256, 0, s, clouds, pfps, N, t1 /*urban*/, t2 /*rest*/);
    dil_l1(t1, 3, 1, t3);               // dilate(urban == 0, 3)
    hipLaunchKernelGGL(k_u8_not, gn, b256, 0, s, t3, N, t1);                          // urban clouds, eroded by 3
    hipLaunchKernelGGL(k_big_small, gpt, b256, 0, s, t2, H, W, t3 /*big*/, far /*small*/);
    dil_l1(t3, 5, 0, t2);
    dil_l1(far, 1, 0, t3);
    hipLaunchKernelGGL(k_or_planes, gn, b256, 0, s, t2, t3, N, far);
    TTC_HIP(c, zero_counts(cnt_a));
    hipLaunchKernelGGL(k_plane_count, gred, b256, 0, s, far, npix, cnt_a);
    near_euclid(far, cnt_a, 3, 9, t2);  // non-urban clouds
    hipLaunchKernelGGL(k_and_planes, gn, b256, 0, s, t2, t1, N, twos);                // value 2 in the reference's sum
    hipLaunchKernelGGL(k_or_planes, gn, b256, 0, s, t2, t1, N, clouds);
    if (debug_stage == 9) return finish(clouds);
    // ---- 10: implausible shadow amounts
    TTC_HIP(c, zero_counts(cnt_a)); TTC_HIP(c, zero_counts(cnt_b));
    hipLaunchKernelGGL(k_mean_flags, gred, b256, 0, s, shadows, npix, cnt_a);
    hipLaunchKernelGGL(k_mean_flags, gred, b256, 0, s, clouds, npix, cnt_b);
    hipLaunchKernelGGL(k_mean_flags, gred, b256, 0, s, twos, npix, cnt_b);            // clouds holds 0 / 1 / 2: mean = (n1 + 2 n2) / N
    hipLaunchKernelGGL(k_shadow_limit, gpt, b256, 0, s, clouds, dem, cnt_a, cnt_b, H, W, shadows);
    if (debug_stage == 10) return finish(shadows);
    hipLaunchKernelGGL(k_or_planes, gn, b256, 0, s, clouds, shadows, N, clouds);
    hipLaunchKernelGGL(k_or_planes, gn, b256, 0, s, fcps, nsr, N, t1);
    TTC_CHECK(dil_l1_3d(t1, 2, d_fcps));
    // ---- 11: false-negative shadows from the per-image blue statistics
    TTC_HIP(c, zero_counts(cnt_b));
    hipLaunchKernelGGL(k_mean_flags, gred, b256, 0, s, clouds, npix, cnt_b);
    hipLaunchKernelGGL(k_mean_flags, gred, b256, 0, s, twos, npix, cnt_b);
    TTC_HIP(c, hipMemsetAsync(acc, 0, sizeof(double) * kMaxT * 6, s));
    hipLaunchKernelGGL(k_inv_blue_moments, gred, b256, 0, s, img, clouds, npix, 0, acc);
    hipLaunchKernelGGL(k_inv_blue_moments, gred, b256, 0, s, img, clouds, npix, 1, acc);
    hipLaunchKernelGGL(k_extra_shadow, gpt, b256, 0, s, img, acc, cnt_b, npix, t1);
    open_planes(t1, 2, 2, t2, t3);
    hipLaunchKernelGGL(k_extra_merge, gpt, b256, 0, s, t3, water, cnt_b, npix, clouds);
    if (debug_stage == 11) return finish(clouds);
    // ---- 12: haze
    TTC_HIP(c, hipMemsetAsync(acc, 0, sizeof(double) * kMaxT * 6, s));
    hipLaunchKernelGGL(k_haze_moments, gred, b256, 0, s, img, clouds, npix, 0, acc);
    hipLaunchKernelGGL(k_haze_moments, gred, b256, 0, s, img, clouds, npix, 1, acc);
    hipLaunchKernelGGL(k_haze_decide, dim3(1), dim3(64), 0, s, acc, T, npix, hazy);
    hipLaunchKernelGGL(k_haze_apply, gpt, b256, 0, s, hazy, npix, clouds);
    hipLaunchKernelGGL(k_u8_to_f32, gn, b256, 0, s, clouds, N, d_clouds);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
