// Cloud / shadow gap-fill: src/preprocessing/cloud_removal.py (CR.py)
//   a6  feather weights           id_areas_to_interp CR.py:774-798, remove_cloud_and_shadows CR.py:910-923
//   a7  aligned cloud-free mosaic make_aligned_mosaic CR.py:578-699
//   a8  per-date NNLS radiometric alignment + blend  align_interp_array_randomforest CR.py:316-575, :943-959
//   a9  clouds left in the mosaic calculate_clouds_in_mosaic CR.py:703-732
//
// All of it is HBM-streaming / small-reduction work: exact EDT by two separable lower-envelope passes
// (only distances <= 12 matter), 20x20 grey closing as separable max / min with scipy's even-size window
// offsets, masked medians by an 8-bit-per-pass radix select, the 11-unknown non-negative least squares
// from Gram matrices (Z'Z with Z = [clipped x | raw x | y], accumulated in double) solved by
// Lawson-Hanson on the host side of the library (microseconds).
//
// Sampling: the reference draws an EVI-stratified sample with the stdlib global RNG.
//   sampler callback != NULL : the library hands the training rows' EVI to the callback and uses the row
//                              indices it returns (the Python mirror replays random.shuffle exactly);
//   sampler callback == NULL : every training row is used with its EXPECTED multiplicity under the
//                              reference's scheme (equal mass per EVI quintile, x10 on the 2 % tails) -- the
//                              deterministic, device-only limit of the reference's estimator.
#include "ttc_internal.h"
#include "radix_select.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kMaxT = 32;

__device__ __forceinline__ int reflect_edge(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }  // scipy 'reflect'

// ------------------------------------------------------------------------------------------------ a6
// The feather weight is b = f(d2) with d2 the squared distance to the nearest masked pixel capped at 169 and
//     f(d2) = 1 - min(sqrt(d2), 12) / 12, values below 0.2 -> 0                                  (CR.py:915-918, float64 like scipy),
// a NON-INCREASING function of the integer d2.  A flat max filter therefore commutes with f as a min filter on d2 (and min
// as max), ties included, so the whole chain -- exact EDT, grey dilation, grey erosion -- runs on ONE BYTE per pixel and f is
// applied once when the weight is stored: bit-identical to filtering float64 planes (round 2: 8 bytes per pixel and pass).
// pass 1: along Y (contiguous): distance to the nearest masked pixel in the row, 13 = none within 12;
// pass 2: along X: d2 = min_dx g(x+dx)^2 + dx^2, capped at 169 (= "further than 12")
// (round 5) four outputs per thread along the scan axis: 28 loads for four outputs instead of up to 25 each (identical results: the first hit
// at increasing distance IS the minimum distance; min over the same candidates)
__global__ void k_edt_rows4(const float* __restrict__ mask, int X, int Y, int clip, unsigned char* __restrict__ g) {
    const int t = blockIdx.y, nb = (Y + 3) / 4;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= X * nb) return;
    const int x = id / nb, y0 = 4 * (id % nb);
    const float* row = mask + ((long)t * X + x) * Y;
    unsigned hit = 0;                                       // bit j: pixel y0 - 12 + j is masked
#pragma unroll
    for (int j = 0; j < 28; ++j) {
        const int yy = y0 - 12 + j;
        if (yy >= 0 && yy < Y) {
            float v = row[yy];
            if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
            if ((1.0f - v) == 0.0f) hit |= 1u << j;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (y0 + r >= Y) break;
        int best = 13;
#pragma unroll
        for (int d = 12; d >= 0; --d)
            if (((hit >> (12 + r - d)) | (hit >> (12 + r + d))) & 1u) best = d;
        g[((long)t * X + x) * Y + y0 + r] = (unsigned char)best;
    }
}
__global__ void k_edt_cols4(const unsigned char* __restrict__ g, int X, int Y, unsigned char* __restrict__ d2) {
    const int t = blockIdx.y;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ((X + 3) / 4) * Y) return;
    const int x0 = 4 * (id / Y), y = id % Y;
    const unsigned char* pl = g + (long)t * X * Y;
    int gv[28];
#pragma unroll
    for (int j = 0; j < 28; ++j) {
        const int xx = x0 - 12 + j;
        gv[j] = (xx >= 0 && xx < X) ? (int)pl[xx * Y + y] : 255;        // out of the plane: never the minimum (255^2 > 169)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (x0 + r >= X) break;
        int best = 169;
#pragma unroll
        for (int dx = -12; dx <= 12; ++dx) {
            const int v = gv[12 + r + dx] * gv[12 + r + dx] + dx * dx;
            best = v < best ? v : best;
        }
        d2[(long)t * X * Y + (long)(x0 + r) * Y + y] = (unsigned char)best;
    }
}
// separable flat filter on the d2 planes with window [lo, hi] and scipy 'reflect' borders.  MAXF refers to the WEIGHT
// (grey dilation of b = min of d2, grey erosion of b = max of d2)
template <bool MAXF, bool ALONG_Y>
__global__ void k_minmax(const unsigned char* __restrict__ in, int X, int Y, int lo, int hi, unsigned char* __restrict__ out) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int x = p / Y, y = p % Y;
    const unsigned char* pl = in + (long)t * X * Y;
    int acc = MAXF ? 255 : 0;
    for (int k = lo; k <= hi; ++k) {
        const int v = ALONG_Y ? pl[x * Y + reflect_edge(y + k, Y)] : pl[reflect_edge(x + k, X) * Y + y];
        acc = MAXF ? min(acc, v) : max(acc, v);
    }
    out[(long)t * X * Y + p] = (unsigned char)acc;
}
// (round 5) the same filter with FOUR outputs per thread along the filter axis: WIN + 3 byte loads for four outputs instead of 4 x WIN -- these
// kernels are bound by their load instructions (20 per output byte), not by bytes.  Identical results (min / max of the same bytes).
template <bool MAXF, bool ALONG_Y, int WIN>
__global__ void k_minmax4(const unsigned char* __restrict__ in, int X, int Y, int lo, unsigned char* __restrict__ out) {
    const int t = blockIdx.y;
    const int nb = ALONG_Y ? (Y + 3) / 4 : (X + 3) / 4;              // blocks of four outputs along the filter axis
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (ALONG_Y ? X * nb : nb * Y)) return;
    const int x = ALONG_Y ? id / nb : 4 * (id / Y), y = ALONG_Y ? 4 * (id % nb) : id % Y;
    const unsigned char* pl = in + (long)t * X * Y;
    int v[WIN + 3];
#pragma unroll
    for (int j = 0; j < WIN + 3; ++j)
        v[j] = ALONG_Y ? pl[x * Y + reflect_edge(min(y + lo + j, 2 * Y - 1), Y)] : pl[reflect_edge(min(x + lo + j, 2 * X - 1), X) * Y + y];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int acc = MAXF ? 255 : 0;
#pragma unroll
        for (int k = 0; k < WIN; ++k) acc = MAXF ? min(acc, v[r + k]) : max(acc, v[r + k]);
        const int xo = ALONG_Y ? x : x + r, yo = ALONG_Y ? y + r : y;
        if (xo < X && yo < Y) out[(long)t * X * Y + (long)xo * Y + yo] = (unsigned char)acc;
    }
}
// dates whose mask is empty keep the (clipped) mask itself (CR.py:786 / :914 `if np.sum(...) > 0`)
__global__ void k_feather_store(const unsigned char* __restrict__ closed, const float* __restrict__ mask, const int* __restrict__ nz,
                                int npix, int clip, float* __restrict__ w) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float m = mask[(long)t * npix + p];
    if (clip) m = fminf(fmaxf(m, 0.f), 1.f);
    double d = sqrt((double)closed[(long)t * npix + p]);
    if (d > 12.0) d = 12.0;
    double v = 1.0 - d / 12.0;
    if (v < 0.2) v = 0.0;
    w[(long)t * npix + p] = nz[t] > 0 ? (float)v : m;
}
__global__ void k_mask_positive(const float* __restrict__ mask, int npix, int clip, int* __restrict__ nz) {
    // (round 5) 16-byte loads where the plane allows and ONE atomic per workgroup: 64 x T workgroups of 4-byte loads with an atomic per wave
    // took 38 us for 18 MB, 256 x T of them 76 us (12 counters under 12 k atomics)
    __shared__ int sc;
    const int t = blockIdx.y;
    if (threadIdx.x == 0) sc = 0;
    __syncthreads();
    int c = 0;
    const float* m = mask + (long)t * npix;
    if ((npix & 3) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
        const float4* m4 = reinterpret_cast<const float4*>(m);
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix / 4; p += gridDim.x * blockDim.x) {
            float4 v = m4[p];
            if (clip) { v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f); v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f); }
            c += (v.x > 0.f) + (v.y > 0.f) + (v.z > 0.f) + (v.w > 0.f);
        }
    } else {
        for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
            float v = m[p];
            if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
            c += v > 0.f;
        }
    }
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&sc, c);
    __syncthreads();
    if (threadIdx.x == 0 && sc) atomicAdd(&nz[t], sc);
}

// ------------------------------------------------------------------------------------------------ a7
template <int TM>
__device__ __forceinline__ void bitonic_sort(float (&a)[TM]) {
#pragma unroll
    for (int k = 2; k <= TM; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float lo = fminf(a[i], a[l]), hi = fmaxf(a[i], a[l]);
                    if ((i & k) == 0) { a[i] = lo; a[l] = hi; } else { a[i] = hi; a[l] = lo; }
                }
            }
}
template <int TM>
__device__ __forceinline__ float median_T(float (&a)[TM], int T) {     // a[t >= T] must be +inf
    bitonic_sort<TM>(a);
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) { if (t == (T - 1) / 2) lo = a[t]; if (t == T / 2) hi = a[t]; }
    return (lo + hi) * 0.5f;
}

// water = median_t NDWI > 0 (any NaN -> not water), NDWI = (G - N) / (G + N)   (CR.py:580-584)
template <int TM, bool OF_MEDIAN>
__global__ void k_water(const float* __restrict__ tiles, int T, int npix, unsigned char* __restrict__ water) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    if (OF_MEDIAN) {     // NDWI of the per-band temporal median (CR.py:936-939)
        float g[TM], n[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            g[t] = t < T ? tiles[((long)t * npix + p) * 10 + 1] : INFINITY;
            n[t] = t < T ? tiles[((long)t * npix + p) * 10 + 3] : INFINITY;
        }
        const float gm = median_T<TM>(g, T), nm = median_T<TM>(n, T);
        water[p] = ((gm - nm) / (gm + nm)) > 0.0f;
    } else {
        float v[TM];
        bool nan = false;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t < T) {
                const float g = tiles[((long)t * npix + p) * 10 + 1], n = tiles[((long)t * npix + p) * 10 + 3];
                v[t] = (g - n) / (g + n);
                nan |= isnan(v[t]);
            } else v[t] = INFINITY;
        }
        water[p] = nan ? 0 : (median_T<TM>(v, T) > 0.0f);
    }
}
// binary dilation with the cross structuring element iterated r times == L1 ball of radius r; `invert` applies
// it to the complement (the reference dilates `1 - x`)
__global__ void k_dilate_diamond(const unsigned char* __restrict__ in, int X, int Y, int r, int invert,
                                 unsigned char* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= X * Y) return;
    const int x = p / Y, y = p % Y;
    bool any = false;
    for (int dx = -r; dx <= r && !any; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= X) continue;
        const int ry = r - abs(dx);
        for (int dy = -ry; dy <= ry; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= Y) continue;
            if ((in[xx * Y + yy] != 0) != (invert != 0)) { any = true; break; }
        }
    }
    out[p] = any;
}

// (round 5) the same diamond (L1 ball) dilation in two separable passes, four outputs per thread: the row pass stores the distance along y to the
// nearest set pixel (r + 1 = none within r), the column pass asks for a row offset dx with |dx| + that distance <= r.  (2 r + 4) / 4 + (2 r + 4) / 4
// loads per output instead of up to 2 r^2 + 2 r + 1 (145 for r = 8).  Same set: exists (dx, dy), |dx| + |dy| <= r, in the plane, set.
constexpr int kDilR = 10;                 // largest radius (the per-thread windows are 2 kDilR + 4 long)
__global__ void k_dil_rows4(const unsigned char* __restrict__ in, int X, int Y, int r, int invert, unsigned char* __restrict__ g) {
    const int nb = (Y + 3) / 4;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= X * nb) return;
    const int x = id / nb, y0 = 4 * (id % nb);
    unsigned hit = 0;                                       // bit j: pixel y0 - r + j is set
#pragma unroll
    for (int j = 0; j < 2 * kDilR + 4; ++j) {
        const int yy = y0 - r + j;
        if (j < 2 * r + 4 && yy >= 0 && yy < Y && ((in[x * Y + yy] != 0) != (invert != 0))) hit |= 1u << j;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (y0 + q >= Y) break;
        int best = r + 1;
        for (int d = r; d >= 0; --d)
            if (((hit >> (r + q - d)) | (hit >> (r + q + d))) & 1u) best = d;
        g[x * Y + y0 + q] = (unsigned char)best;
    }
}
__global__ void k_dil_cols4(const unsigned char* __restrict__ g, int X, int Y, int r, unsigned char* __restrict__ out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= ((X + 3) / 4) * Y) return;
    const int x0 = 4 * (id / Y), y = id % Y;
    int gv[2 * kDilR + 4];
#pragma unroll
    for (int j = 0; j < 2 * kDilR + 4; ++j) {
        const int xx = x0 - r + j;
        gv[j] = (j < 2 * r + 4 && xx >= 0 && xx < X) ? (int)g[xx * Y + y] : 255;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (x0 + q >= X) break;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 2 * kDilR + 1; ++j) {          // dx = j - r
            const int dx = j - r;
            if (j <= 2 * r) any |= gv[q + j] + (dx < 0 ? -dx : dx) <= r;
        }
        out[(x0 + q) * Y + y] = any;
    }
}
// per date i: valid = (w_i < 0.25) & land & (some other date b with w_b < 1); ref = mean of those dates
__global__ void k_mosaic_ref(const float* __restrict__ tiles, const float* __restrict__ w, const unsigned char* __restrict__ water,
                             int T, int npix, int i, float* __restrict__ ref, unsigned char* __restrict__ valid,
                             int* __restrict__ count) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int ok = 0;
    if (p < npix) {
        if (w[(long)i * npix + p] < 0.25f && !water[p]) {
            float s[10];
#pragma unroll
            for (int c = 0; c < 10; ++c) s[c] = 0.f;
            int n = 0;
            for (int b = 0; b < T; ++b) {
                if (b == i || !(w[(long)b * npix + p] < 1.0f)) continue;
                const float* v = tiles + ((long)b * npix + p) * 10;
#pragma unroll
                for (int c = 0; c < 10; ++c) s[c] += v[c];
                ++n;
            }
            if (n > 0) {
                ok = 1;
#pragma unroll
                for (int c = 0; c < 10; ++c) ref[(long)p * 10 + c] = s[c] / (float)n;
            }
        }
        valid[p] = ok;
    }
    int c = ok;
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// ---- radix select: k-th smallest of the valid elements, 8 bits per pass, batched over problems -------------
using namespace ttcsel;
struct DatePlan { int proceed, t0, nt, nrows, fitted; };     // device-resident control block of one gap-fill date

// selection sources: value of problem q at element p (false = element not in the set)
struct SrcMosaic {      // q = band*4 + which*2 + rank ; which 0 -> ref[p][band], 1 -> src[p][band]
    const float* ref; const float* src; const unsigned char* valid; int npix;
    __device__ int count() const { return npix; }
    __device__ bool get(int q, int p, float& v) const {
        if (!valid[p]) return false;
        v = (((q >> 1) & 1) ? src : ref)[(long)p * 10 + (q >> 2)];
        return true;
    }
};
struct SrcBlueRed {     // q = band2*2 + rank ; band2 0 -> mosaic blue, 1 -> mosaic red ; set = ~only1
    const float* mosaic; const unsigned char* only1; int npix;
    __device__ int count() const { return npix; }
    __device__ bool get(int q, int p, float& v) const {
        if (only1[p]) return false;
        v = mosaic[(long)p * 10 + ((q >> 1) ? 2 : 0)];
        return true;
    }
};

// sum (pass 0) / sum of squared deviations from sum/n (pass 1) of the valid elements, 20 problems = (band, which)
__global__ void k_stat_sum(SrcMosaic s, const double* __restrict__ sum, const int* __restrict__ count, double* __restrict__ out) {
    const int q = blockIdx.y;
    const double m = sum ? sum[q] / (double)max(*count, 1) : 0.0;
    double acc = 0.0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < s.npix; p += gridDim.x * blockDim.x) {
        float v;
        if (!s.get(((q >> 1) << 2) | ((q & 1) << 1), p, v)) continue;
        const double d = (double)v - m;
        acc += sum ? d * d : d;
    }
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[q], acc);
}

// gain / offset per band from (median, std) of ref and src; ok flag (CR.py:624-641)
struct AlignPar { float k[10], add[10]; int ok; int any_land; };
__global__ void k_align_params(const SelState* __restrict__ st, const double* __restrict__ var, const int* __restrict__ count,
                               const int* __restrict__ n_land, AlignPar* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const int n = *count;
    AlignPar ap;
    ap.ok = n > 1000;
    ap.any_land = *n_land > 0;
    for (int b = 0; b < 10; ++b) {
        // problems q = band*4 + which*2 + rank ; median = mean of the two middle order statistics
        const float mr = 0.5f * (fkey_inv(st[b * 4 + 0].prefix) + fkey_inv(st[b * 4 + 1].prefix));
        const float ms = 0.5f * (fkey_inv(st[b * 4 + 2].prefix) + fkey_inv(st[b * 4 + 3].prefix));
        const float sr = (float)sqrt(var[b * 2 + 0] / (double)n), ss = (float)sqrt(var[b * 2 + 1] / (double)n);
        const float k = sr / ss;
        ap.k[b] = k; ap.add[b] = mr - ms * k;
    }
    *out = ap;
}
__global__ void k_mosaic_accum(const float* __restrict__ tiles, float* __restrict__ w, const unsigned char* __restrict__ water,
                               const AlignPar* __restrict__ app, int npix, int i, float* __restrict__ mosaic) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const AlignPar& ap = *app;
    if (ap.ok) {
        const float wi = 1.0f - w[(long)i * npix + p];
        const float* v = tiles + ((long)i * npix + p) * 10;
        const bool land = !water[p];
#pragma unroll
        for (int c = 0; c < 10; ++c) {
            const float a = land ? v[c] * ap.k[c] + ap.add[c] : v[c];
            mosaic[(long)p * 10 + c] = mosaic[(long)p * 10 + c] + wi * a;
        }
    } else if (ap.any_land) {
        w[(long)i * npix + p] = 1.0f;                          // CR.py:679-680
    }
}
// mosaic / divisor, NaN -> 10th percentile over dates, clamp to [min_t, max_t] (CR.py:683-689)
template <int TM>
__global__ void k_mosaic_final(const float* __restrict__ tiles, const float* __restrict__ divisor, int T, int npix,
                               float* __restrict__ mosaic) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float d = divisor[p];
    if (d < 0.f) d = 0.f;
    for (int c = 0; c < 10; ++c) {
        float v[TM];
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            v[t] = t < T ? tiles[((long)t * npix + p) * 10 + c] : INFINITY;
            if (t < T) { mn = fminf(mn, v[t]); mx = fmaxf(mx, v[t]); }
        }
        float m = mosaic[(long)p * 10 + c] / d;
        if (isnan(m)) {
            bitonic_sort<TM>(v);
            const double pos = 0.1 * (T - 1);
            const int lo = (int)floor(pos);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int t = 0; t < TM; ++t) { if (t == lo) a = v[t]; if (t == lo + 1 && lo + 1 < T) b = v[t]; }
            if (lo + 1 >= T) b = a;
            m = (float)((double)a + ((double)b - (double)a) * (pos - lo));
        }
        m = fmaxf(m, mn);
        m = fminf(m, mx);
        mosaic[(long)p * 10 + c] = m;
    }
}
__global__ void k_divisor(const float* __restrict__ w, int T, int npix, float* __restrict__ divisor) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += 1.0f - w[(long)t * npix + p];
    divisor[p] = s;
}
__global__ void k_count_land(const unsigned char* __restrict__ water, int npix, int* __restrict__ n_land) {
    int c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) c += !water[p];
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_land, c);
}

// ------------------------------------------------------------------------------------------------ a8
__device__ __forceinline__ float snow_prob_px(const float* v) {           // CR.py:348-370
#pragma clang fp contract(off)
    float ndsi = (v[1] - v[8]) / (v[1] + v[8]);
    if (ndsi < 0.10f) ndsi = 0.f;                      // NaN compares false: stays NaN, like numpy
    if (ndsi > 0.42f) ndsi = 0.42f;
    float p = (ndsi - 0.1f) / 0.32f;
    if (v[3] < 0.10f) p = 0.f;
    if (v[3] > 0.35f && p > 0.f) p = 1.f;
    if (v[0] < 0.10f) p = 0.f;
    if (v[0] > 0.22f && p > 0.f) p = 1.f;
    if ((v[0] / v[2]) < 0.75f) p = 0.f;
    return p;
}
// The mean snow probability is re-evaluated for every date because the stack is blended in place (CR.py:372); only ONE
// date changes between evaluations, so the per-date probabilities are cached and the mean re-sums T floats per pixel
// (same values, same order) instead of re-reading the whole stack.
// screen (single-call tile path, else nullptr) [T][2]: per date, the two counts behind process_tile's date screening that
// this pass sees for free -- pixels with snow_filter(...) true (job.py:799-818: the same ramp, as a flag) and "missing" pixels
// (id_missing_px, interpolation.py:5-23: more than one band == 0 or >= 1)
__global__ void k_snow_prob_all(const float* __restrict__ tiles, int npix, float* __restrict__ snowp, int* __restrict__ screen) {
    const int t = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int sn = 0, miss = 0;
    if (p < npix) {
        const float* v = tiles + ((long)t * npix + p) * 10;
        const float pr = snow_prob_px(v);
        snowp[(long)t * npix + p] = pr;
        if (screen) {
            int bad = 0;
#pragma unroll
            for (int b = 0; b < 10; ++b) bad += (v[b] == 0.0f) + (v[b] >= 1.0f);
            sn = pr > 0.f; miss = bad > 1;
        }
    }
    if (screen) {
        for (int k = 32; k >= 1; k >>= 1) { sn += __shfl_xor(sn, k); miss += __shfl_xor(miss, k); }
        if ((threadIdx.x & 63) == 0) { if (sn) atomicAdd(&screen[2 * t], sn); if (miss) atomicAdd(&screen[2 * t + 1], miss); }
    }
}
// ONE pass over the stack for the three per-pixel products the gap-fill takes from it before its date loop (round 6).  Before: k_water<false>
// (aligned mosaic, CR.py:580-584), k_water<true> (CR.py:936-939) and k_snow_prob_all each fetched all T x 40-byte records of every pixel for 8
// to 40 of their bytes -- 3 x 183 MB per T = 12 tile (profiles/r06_pmc_preprocess_total.json).  Same expressions, same order per output:
// bit-identical masks and probabilities.  Thread = pixel; the per-date screening counts go through LDS (2 T global atomics per workgroup).
template <int TM>
__global__ __launch_bounds__(256) void k_water_snow(const float* __restrict__ tiles, int T, int npix, unsigned char* __restrict__ water_ndwi,
                                                    unsigned char* __restrict__ water_med, float* __restrict__ snowp, int* __restrict__ screen) {
#pragma clang fp contract(off)
    __shared__ int sc[2 * TM];
    if (threadIdx.x < 2 * TM) sc[threadIdx.x] = 0;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < npix;
    float g[TM], n[TM], v[TM];
    bool nan = false;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        g[t] = INFINITY; n[t] = INFINITY; v[t] = INFINITY;
        if (t < T) {
            int sn = 0, miss = 0;
            if (live) {
                const float2* src = reinterpret_cast<const float2*>(tiles + ((long)t * npix + p) * 10);
                float r[10];
#pragma unroll
                for (int b = 0; b < 5; ++b) { const float2 u = src[b]; r[2 * b] = u.x; r[2 * b + 1] = u.y; }
                g[t] = r[1]; n[t] = r[3];
                v[t] = (r[1] - r[3]) / (r[1] + r[3]);
                nan |= isnan(v[t]);
                const float pr = snow_prob_px(r);
                snowp[(long)t * npix + p] = pr;
                if (screen) {
                    int bad = 0;
#pragma unroll
                    for (int b = 0; b < 10; ++b) bad += (r[b] == 0.0f) + (r[b] >= 1.0f);
                    sn = pr > 0.f; miss = bad > 1;
                }
            }
            if (screen) {
                for (int k = 32; k >= 1; k >>= 1) { sn += __shfl_xor(sn, k); miss += __shfl_xor(miss, k); }
                if ((threadIdx.x & 63) == 0) { if (sn) atomicAdd(&sc[2 * t], sn); if (miss) atomicAdd(&sc[2 * t + 1], miss); }
            }
        }
    }
    if (live) {
        water_ndwi[p] = nan ? 0 : (median_T<TM>(v, T) > 0.0f);                 // k_water<TM, false>
        const float gm = median_T<TM>(g, T), nm = median_T<TM>(n, T);          // k_water<TM, true>
        water_med[p] = ((gm - nm) / (gm + nm)) > 0.0f;
    }
    if (screen) {
        __syncthreads();
        if (threadIdx.x < 2 * T && sc[threadIdx.x]) atomicAdd(&screen[threadIdx.x], sc[threadIdx.x]);
    }
}
__global__ void k_snow_mean_cached(const float* __restrict__ snowp, int T, int npix, float* __restrict__ snow) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += snowp[(long)t * npix + p];
    snow[p] = s / (float)T;
}
__device__ __forceinline__ float evi_unclipped(const float* v) {          // CR.py:332-345
#pragma clang fp contract(off)
    const float e = 2.5f * ((v[3] - v[2]) / (((v[3] + (6.0f * v[2])) - (7.5f * v[0])) + 1.0f));
    return fminf(fmaxf(e, -1.5f), 1.5f);
}
// all dates at once (the feather weights and the water mask do not change inside the date loop)
__global__ void k_date_counts_all(const float* __restrict__ w, int npix, int* __restrict__ out /*[T][4]*/) {
    const int date = blockIdx.y;
    int a = 0, b = 0, c = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const float v = w[(long)date * npix + p];
        a += v > 0.f; b += v == 0.f; c += v < 1.f;
    }
    for (int k = 32; k >= 1; k >>= 1) { a += __shfl_xor(a, k); b += __shfl_xor(b, k); c += __shfl_xor(c, k); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[date * 4 + 0], a); atomicAdd(&out[date * 4 + 1], b); atomicAdd(&out[date * 4 + 2], c); }
}
// training rows of dates [t0, t1): pixels with w_t == 0 and not water, in (t, pixel) order -- 2-level scan compaction
__global__ void k_rows_count(const float* __restrict__ w, const unsigned char* __restrict__ water, int npix, int t0, int nt,
                             const DatePlan* __restrict__ plan, int* __restrict__ blk) {
    if (plan) { plan += blockIdx.y; blk += (long)blockIdx.y * gridDim.x; t0 = plan->t0; nt = plan->nt; }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int f = 0;
    if (i < (long)nt * npix) { const int t = t0 + (int)(i / npix), p = (int)(i % npix); f = (w[(long)t * npix + p] == 0.f) && !water[p]; }
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
__global__ void k_rows_scan(int* __restrict__ blk, int nblk, int* __restrict__ total, int total_stride) {   // one block per list
    blk += (long)blockIdx.x * nblk; total += (long)blockIdx.x * total_stride;
    __shared__ int carry;
    __shared__ int tmp[1024];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blk[i] : 0;
        tmp[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int a = threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
            __syncthreads();
            tmp[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < nblk) blk[i] = carry + tmp[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += tmp[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void k_rows_fill(const float* __restrict__ w, const unsigned char* __restrict__ water, const float* __restrict__ tiles,
                            int npix, int t0, int nt, const DatePlan* __restrict__ plan, const int* __restrict__ blk,
                            int* __restrict__ rows, float* __restrict__ evi) {
    if (plan) {
        plan += blockIdx.y; blk += (long)blockIdx.y * gridDim.x; rows += (long)blockIdx.y * 3 * npix; t0 = plan->t0; nt = plan->nt;
        if (evi) evi += (long)blockIdx.y * 3 * npix;
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int f = 0, t = 0, p = 0;
    if (i < (long)nt * npix) { t = t0 + (int)(i / npix); p = (int)(i % npix); f = (w[(long)t * npix + p] == 0.f) && !water[p]; }
    __shared__ int wbase[4];
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wbase[wv] = __popcll(m);
    __syncthreads();
    int off = blk[blockIdx.x];
    for (int k = 0; k < wv; ++k) off += wbase[k];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (f) { rows[off] = (int)((long)(t - t0) * npix + p); if (evi) evi[off] = evi_unclipped(tiles + ((long)t * npix + p) * 10); }
}
// EVI of the listed rows from the CURRENT stack (the list itself is date-loop invariant, the values are not)
__global__ void k_rows_evi(const int* __restrict__ rows, const DatePlan* __restrict__ plan, const float* __restrict__ tiles, int npix,
                           float* __restrict__ evi) {
    const int n = plan->nrows, t0 = plan->t0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int rr = rows[i];
        evi[i] = evi_unclipped(tiles + ((long)(t0 + rr / npix) * npix + rr % npix) * 10);
    }
}

// Z'Z for Z = [clip(x,0.005,1)(10) snow | x(10) snow | y(10)] (32 columns), rows given by an index list with
// optional per-row weights; double accumulation; one 32x32 partial per block, reduced by k_gram_reduce.
struct GramArgs {
    const float* tiles; const float* mosaic; const float* snow; const int* rows; const int* sample; const float* weight;
    long nsample; int npix; int t0;
    const DatePlan* plan;     // device-controlled launch: nsample = plan->nrows, t0 = plan->t0
};
__global__ __launch_bounds__(256) void k_gram(GramArgs a, double* __restrict__ partial) {
    // 256 rows per step: every thread gathers ONE row (the gather is a chain of dependent scattered loads -- with 64
    // gathering threads per step it was the whole kernel), then owns Z'Z[r0][c0..c0+3] over the staged rows.
    constexpr int R = 256;
    __shared__ float zs[R][33];
    __shared__ float ws[R];
    double acc[4] = {0, 0, 0, 0};
    if (a.plan) { a.nsample = a.plan->nrows; a.t0 = a.plan->t0; }
    const int tid = threadIdx.x;
    const int r0 = tid >> 3, c0 = (tid & 7) * 4;
    for (long base = (long)blockIdx.x * R; base < a.nsample; base += (long)gridDim.x * R) {
        __syncthreads();
        {
            const long s = base + tid;
            float wgt = 0.f;
            if (s < a.nsample) {
                const int row = a.sample ? a.sample[s] : (int)s;
                const int rr = a.rows[row];
                const int t = a.t0 + rr / a.npix, p = rr % a.npix;
                const float* x = a.mosaic + (long)p * 10;
                const float* y = a.tiles + ((long)t * a.npix + p) * 10;
                const float sn = a.snow[p];
#pragma unroll
                for (int c = 0; c < 10; ++c) { const float xv = x[c]; zs[tid][c] = fminf(fmaxf(xv, 0.005f), 1.0f); zs[tid][11 + c] = xv; zs[tid][22 + c] = y[c]; }
                zs[tid][10] = sn; zs[tid][21] = sn;
                wgt = a.weight ? a.weight[row] : 1.0f;
            } else {
#pragma unroll
                for (int c = 0; c < 32; ++c) zs[tid][c] = 0.f;
            }
            ws[tid] = wgt;
        }
        __syncthreads();
        const int n = (int)min((long)R, a.nsample - base);
        for (int s = 0; s < n; ++s) {
            const double zr = (double)zs[s][r0] * (double)ws[s];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += zr * (double)zs[s][c0 + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) partial[(long)blockIdx.x * 1024 + r0 * 32 + c0 + k] = acc[k];
}
__global__ void k_gram_reduce(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
    // 16 Z'Z entries per workgroup (coalesced along the entry index), 16 lanes split the blocks of each entry
    __shared__ double red[16][17];
    const int e = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    double s = 0.0;
    for (int b = part; b < nblk; b += 16) s += partial[(long)b * 1024 + i];
    red[part][e] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][e];      // fixed order: deterministic
        out[i] = t;
    }
}
// prediction + blend (CR.py:561-569, :954-955): pixels with w_d > 0 get [fill, snow] . beta, then
// tile = tile * (1 - w) + pred * w
struct Beta { double b[10][11]; int fitted; };
__global__ void k_predict_blend(float* __restrict__ tiles, const float* __restrict__ w, const float* __restrict__ mosaic,
                                float* __restrict__ snow, const Beta* __restrict__ bep, int npix, int date,
                                float* __restrict__ snowp, int T = 0, int clip01 = 0) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float wd = w[(long)date * npix + p];
    if (!(wd > 0.f)) return;                                   // tile * 1 + 0 * 0: unchanged
    float* tv = tiles + ((long)date * npix + p) * 10;
    const float* mv = mosaic + (long)p * 10;
    // the mean snow probability as it stands before this date is blended (CR.py:372): `snow` holds it in both forms of the date
    // loop -- the batched form (T > 0) keeps it current below, the replayed-sampler form refreshes it with k_snow_mean_cached
    const float snf = snow[p];
    const double sn = (double)snf;
    const Beta& be = *bep;
    float bl[10];
    for (int c = 0; c < 10; ++c) {
        float pred = mv[c];
        if (be.fitted) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 10; ++j) s += (double)mv[j] * be.b[c][j];
            s += sn * be.b[c][10];
            pred = (float)s;
        }
        bl[c] = tv[c] * (1.0f - wd) + pred * wd;
    }
    snowp[(long)date * npix + p] = snow_prob_px(bl);           // keep the cache of this date current (unclipped values)
    if (T > 0) {                                                // ... and the mean: same float sum in date order as k_snow_mean_cached
        float sm = 0.f;
        for (int t = 0; t < T; ++t) sm += snowp[(long)t * npix + p];
        snow[p] = sm / (float)T;
    }
    // clip01 (single-call tile path): process_tile's final np.clip(sentinel2, 0, 1) (job.py:993) applied at the only place a
    // value can leave [0, 1] -- decoded uint16 / 65535 and their bilinear means cannot, an NNLS prediction can.  Nothing in
    // the rest of the gap-fill reads a blended pixel (training rows have w == 0), so clipping here equals clipping at the end.
    for (int c = 0; c < 10; ++c) tv[c] = clip01 ? fminf(fmaxf(bl[c], 0.f), 1.f) : bl[c];
}

// ------------------------------------------------------------------------------------------------ a9
__global__ void k_only1(const float* __restrict__ w, const unsigned char* __restrict__ pf_dil, int T, int npix,
                        unsigned char* __restrict__ only1, int* __restrict__ n_only) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int o = 0;
    if (p < npix) {
        int clear = 0;
        for (int t = 0; t < T; ++t) clear += !(w[(long)t * npix + p] > 0.f);
        o = (clear < 2) || pf_dil[p];
        only1[p] = o;
    }
    int c = o;
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_only, c);
}
__global__ void k_cloud_flags(const float* __restrict__ mosaic, const unsigned char* __restrict__ only1,
                              const unsigned char* __restrict__ pf_dil, const float* __restrict__ thr, int npix,
                              unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float* m = mosaic + (long)p * 10;
    bool c = (m[0] > thr[0]) && (m[2] > thr[1]) && only1[p] && (((m[0] + m[1]) + m[2]) < 1.0f);
    if (pf_dil[p]) c = false;
    out[p] = c;
}
__global__ void k_add_clouds(float* __restrict__ w, const unsigned char* __restrict__ clouds, int T, int npix) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix || !clouds[p]) return;
    for (int t = 0; t < T; ++t) { float v = w[(long)t * npix + p] + 1.0f; w[(long)t * npix + p] = v > 1.f ? 1.f : v; }
}
// ---- device-side control of the per-date fit (no host round trips) --------------------------------
// screen / spec (single-call tile path, else nullptr): the date-dropping rules of process_tile that the single call cannot
// take (they change T and re-run the detection) are evaluated here and reported in spec[3]:
//   1  a date has >= X^2 / 2 missing pixels                      (id_missing_px(sentinel2, 2), job.py:786)
//   2  more than 10 dates are > 25 % snow                         (job.py:822-824)
//   4  the feather weights of a date cover > 90 % of the tile     (job.py:866 and the two repeats; tested on the closing-20
//      weights, which are >= the closing-15 weights of id_areas_to_interp pixel by pixel: conservative)
__global__ void k_date_plan(const int* __restrict__ counters, int npix, int X, int T, DatePlan* __restrict__ plans,
                            int* __restrict__ remove_flags, const int* __restrict__ screen, int* __restrict__ spec) {
    const int date = threadIdx.x;
    if (spec && date == 0) {
        int bits = 0, snowy = 0;
        for (int t = 0; t < T; ++t) {
            snowy += ((double)screen[2 * t] / (double)npix) > 0.25;
            if ((double)screen[2 * t + 1] >= ((double)X * (double)X) / 2.0) bits |= 1;
            if (((double)counters[4 * t] / (double)npix) > 0.9) bits |= 4;
        }
        if (snowy > 10) bits |= 2;
        spec[3] = bits;
    }
    if (date >= T) return;
    counters += date * 4;
    DatePlan* plan = plans + date;
    const int c0 = counters[0], c1 = counters[1], c2 = counters[2];
    DatePlan p;
    p.proceed = c0 > 0 && c1 > 0 && ((double)c2 / npix) > 0.01;                 // CR.py:377-378
    if (c1 > 40000) { p.t0 = date; p.nt = 1; }                                     // CR.py:394-402
    else {
        const int a = max(date == T - 1 ? date - 2 : date - 1, 0), b = min(date + 2, T);
        p.t0 = a; p.nt = b - a;
    }
    if (!p.proceed) p.nt = 0;
    p.nrows = 0; p.fitted = 0;
    *plan = p;
    remove_flags[date] = (c2 == 0);                                                // CR.py:958-959
}
struct StrataDev { float b[6]; int cnt[5]; };
__device__ __forceinline__ int stratum_of(float e, const float* b) { return e < b[1] ? 0 : (e < b[2] ? 1 : (e < b[3] ? 2 : (e < b[4] ? 3 : 4))); }

// ---- date-batched form of the per-date fit (device sampler) ------------------------------------------------------
// A training row is a pixel with w_t == 0, and the blend only ever writes pixels with w_d > 0 of date d: the row
// VALUES (targets, EVI, strata, weights) never change inside the date loop.  The only quantity that does is the mean
// snow probability (CR.py:372), one of the 11 regressors.  So everything but the snow row / column of Z'Z is formed for
// all dates in batched launches; a date then costs three launches: the snow products over its rows, the NNLS (which
// first completes Z'Z), predict + blend.
__global__ void k_sel_init_evi(SelState* __restrict__ st, const DatePlan* __restrict__ plans, int T, PctList pl) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= T * 12) return;
    const long long n = plans[q / 12].nrows;
    const int j = q % 12;
    SelState ss; ss.prefix = 0; ss.mask = 0; ss.k = 0;
    if (n > 0) {
        const double pos = pl.pct[j >> 1] / 100.0 * (double)(n - 1);
        const long long lo = (long long)floor(pos) + (j & 1);
        ss.k = lo > n - 1 ? n - 1 : lo;
    }
    st[q] = ss;
}
// one radix-select pass of the 12 percentile problems of date blockIdx.y: the EVI list is read once for all 12
__global__ __launch_bounds__(256) void k_evi_hist(const float* __restrict__ evi_all, const DatePlan* __restrict__ plans, int npix,
                                                   const SelState* __restrict__ st, int shift, unsigned* __restrict__ hist) {
    __shared__ unsigned h[12 * 256];
    __shared__ unsigned pf[12], mk[12];
    __shared__ int same[12];                                   // first problem with the same prefix: they share one histogram
    const int d = blockIdx.y;
    for (int k = threadIdx.x; k < 12 * 256; k += blockDim.x) h[k] = 0;
    if (threadIdx.x < 12) { pf[threadIdx.x] = st[d * 12 + threadIdx.x].prefix; mk[threadIdx.x] = st[d * 12 + threadIdx.x].mask; }
    __syncthreads();
    if (threadIdx.x < 12) {
        int r = threadIdx.x;
        for (int q = threadIdx.x - 1; q >= 0; --q) if (pf[q] == pf[threadIdx.x] && mk[q] == mk[threadIdx.x]) r = q;
        same[threadIdx.x] = r;
    }
    __syncthreads();
    const int n = plans[d].nrows;
    const float* evi = evi_all + (long)d * 3 * npix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned k = fkey(evi[i]);
        const unsigned bin = (k >> shift) & 255u;
#pragma unroll
        for (int q = 0; q < 12; ++q)
            if (same[q] == q && (k & mk[q]) == pf[q]) atomicAdd(&h[q * 256 + bin], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 12 * 256; k += blockDim.x) {
        const unsigned v = h[same[k >> 8] * 256 + (k & 255)];
        if (v) atomicAdd(&hist[(long)d * 12 * 256 + k], v);
    }
}
__global__ void k_strata_thresholds_all(const SelState* __restrict__ st, const DatePlan* __restrict__ plans, int T, PctList pl,
                                        StrataDev* __restrict__ sd) {
    const int d = threadIdx.x;
    if (d >= T) return;
    const long long n = plans[d].nrows;
    for (int k = 0; k < 6; ++k) {
        const double pos = n > 0 ? pl.pct[k] / 100.0 * (double)(n - 1) : 0.0;
        const double a = fkey_inv(st[d * 12 + 2 * k].prefix), b = fkey_inv(st[d * 12 + 2 * k + 1].prefix);
        sd[d].b[k] = (float)(a + (b - a) * (pos - floor(pos)));
    }
    for (int k = 0; k < 5; ++k) sd[d].cnt[k] = 0;
}
__global__ void k_strata_count_all(const float* __restrict__ evi_all, const DatePlan* __restrict__ plans, int npix, StrataDev* __restrict__ sd) {
    __shared__ int c[5];
    const int d = blockIdx.y;
    if (threadIdx.x < 5) c[threadIdx.x] = 0;
    __syncthreads();
    const int n = plans[d].nrows;
    const float* evi = evi_all + (long)d * 3 * npix;
    int mine[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int sidx = stratum_of(evi[i], sd[d].b);
#pragma unroll
        for (int k = 0; k < 5; ++k) mine[k] += sidx == k;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int v = mine[k];
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&c[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 5 && c[threadIdx.x]) atomicAdd(&sd[d].cnt[threadIdx.x], c[threadIdx.x]);
}
__global__ void k_row_weights_all(const float* __restrict__ evi_all, const DatePlan* __restrict__ plans, int npix,
                                  const StrataDev* __restrict__ sdv, float* __restrict__ weight_all) {
    const int d = blockIdx.y;
    const int n = plans[d].nrows;
    const StrataDev* sd = sdv + d;
    const float* evi = evi_all + (long)d * 3 * npix;
    float* weight = weight_all + (long)d * 3 * npix;
    const double n_i = (double)(min(90000, n) / 5);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float e = evi[i];
        const int cn = sd->cnt[stratum_of(e, sd->b)];
        float w = cn > 0 ? (float)fmin(1.0, n_i / (double)cn) : 0.f;
        if (e < sd->b[0]) w += 10.f;
        if (e >= sd->b[5]) w += 10.f;
        weight[i] = w;
    }
}
// Z'Z without the snow regressor for date blockIdx.y (columns 10 and 21 are left zero; k_gram_snow supplies them), on the
// fp64 matrix cores: G[i][j] += sum_k (w_k z_k[i]) * z_k[j] is a 32 x 32 x rows GEMM, four rows per v_mfma_f64_16x16x4_f64 and
// three 16 x 16 blocks (G00, G01, G11; G10 = G01').  Same products in double as the LDS-tiled vector form it replaces
// ((double)z_i * (double)w, times (double)z_j, FMA-accumulated), 3 matrix instructions per 4 rows instead of 1024 DFMA lanes
// with two LDS operand reads each (round 2: 305 us per tile).  A lane (c = lane & 15, k = lane >> 4) supplies row k of the
// step, columns c (block 0) and 16 + c (block 1): two gathered floats.  Row indices / weights of 64 consecutive rows arrive
// as one coalesced load each per wave and are handed round with ds_bpermute.
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gram_all(const float* __restrict__ tiles, const float* __restrict__ mosaic,
                                                   const int* __restrict__ rows_all, const float* __restrict__ weight_all,
                                                   const DatePlan* __restrict__ plans, int npix, double* __restrict__ partial) {
#pragma clang fp contract(off)
    __shared__ double red[4][3][16][16];
    const int d = blockIdx.y;
    const int nsample = plans[d].nrows, t0 = plans[d].t0;
    const int* rows = rows_all + (long)d * 3 * npix;
    const float* weight = weight_all + (long)d * 3 * npix;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, k = lane >> 4;
    // block 0 column c: 0..9 clip(x_c), 10 zero (snow), 11..15 x_{c-11};  block 1 column 16 + c: 16..20 x_{c+5}, 21 zero, 22..31 y_{c-6}
    const bool a_zero = c == 10, a_clip = c < 10;
    const int a_ch = c < 10 ? c : (c > 10 ? c - 11 : 0);
    const bool b_zero = c == 5, b_y = c >= 6;
    const int b_ch = c < 5 ? c + 5 : (c >= 6 ? c - 6 : 0);
    f64x4 g00 = {0, 0, 0, 0}, g01 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
    const int chunk0 = blockIdx.x * 4 + wv, nchunks = gridDim.x * 4;
    for (int base = chunk0 * 64; base < nsample; base += nchunks * 64) {
        const int mine = base + lane;
        const int rr_l = mine < nsample ? rows[mine] : 0;
        const float w_l = mine < nsample ? weight[mine] : 0.f;          // rows past the end: weight 0 -> contribute nothing
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            const int src = 4 * st + k;
            const int rr = __shfl(rr_l, src);
            const float w = __shfl(w_l, src);
            const int dt = (rr >= npix) + (rr >= 2 * npix);
            const int p = rr - dt * npix;
            const float* x = mosaic + (long)p * 10;
            const float* y = tiles + ((long)(t0 + dt) * npix + p) * 10;
            float za = x[a_ch];
            if (a_clip) za = fminf(fmaxf(za, 0.005f), 1.0f);
            if (a_zero) za = 0.f;
            float zb = b_y ? y[b_ch] : x[b_ch];
            if (b_zero) zb = 0.f;
            const double wa = (double)za * (double)w, wb = (double)zb * (double)w;
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, (double)za, g00, 0, 0, 0);
            g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, (double)zb, g01, 0, 0, 0);
            g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(wb, (double)zb, g11, 0, 0, 0);
        }
    }
    // C / D layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wv][0][k + 4 * r][c] = g00[r];
        red[wv][1][k + 4 * r][c] = g01[r];
        red[wv][2][k + 4 * r][c] = g11[r];
    }
    __syncthreads();
    // the four waves' partials in a fixed order -> this block's 32 x 32 partial (symmetric fill)
    double* out = partial + ((long)d * gridDim.x + blockIdx.x) * 1024;
    for (int e = threadIdx.x; e < 768; e += 256) {
        const int blk = e >> 8, i = (e >> 4) & 15, j2 = e & 15;
        const double v = ((red[0][blk][i][j2] + red[1][blk][i][j2]) + red[2][blk][i][j2]) + red[3][blk][i][j2];
        if (blk == 0) out[i * 32 + j2] = v;
        else if (blk == 2) out[(16 + i) * 32 + 16 + j2] = v;
        else { out[i * 32 + 16 + j2] = v; out[(16 + j2) * 32 + i] = v; }
    }
}
__global__ void k_gram_reduce_all(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
    __shared__ double red[16][17];
    const int e = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    partial += (long)blockIdx.y * nblk * 1024; out += (long)blockIdx.y * 1024;
    double s = 0.0;
    for (int b = part; b < nblk; b += 16) s += partial[(long)b * 1024 + i];
    red[part][e] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][e];
        out[i] = t;
    }
}
// snow products of ONE date over its rows: sv[i] = sum_rows w * z_i * snow (z_10 = z_21 = snow), snow = the CURRENT mean of
// the per-date probabilities (`snowm`, kept exact by k_predict_blend: the same float sum in date order as k_snow_mean_cached).
// Lane = COLUMN: the 32 lanes of a half-wave own the 32 columns of Z and walk the row list together, two rows per wave and
// step -- a lane loads ONE float per row (its own x / y element; row index, weight and snow mean are half-wave broadcasts)
// and accumulates its column in double, so there is no cross-lane reduction per row.  (Round 2: a thread per row with 32
// double accumulators and a 32 x 6-step shuffle tree for 2-3 rows each -- 65 us per date, 0.78 ms of a tile's 5.2 ms.)
constexpr int kSnowBlocks = 256;       // x 1024 threads; k_nnls sums the per-block partials in a fixed order (bit-reproducible)
__global__ __launch_bounds__(1024) void k_gram_snow(const float* __restrict__ tiles, const float* __restrict__ mosaic,
                                                     const float* __restrict__ snowm, const int* __restrict__ rows,
                                                     const float* __restrict__ weight, const DatePlan* __restrict__ plan,
                                                     int npix, double* __restrict__ partial /*[kSnowBlocks][32]*/) {
#pragma clang fp contract(off)
    __shared__ double red[16][32];
    const int nsample = plan->nrows, t0 = plan->t0;
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    // column -> source: 0..9 clip(x_c), 10 snow, 11..20 x_c, 21 snow, 22..31 y_c
    const bool is_y = col >= 22, is_sn = col == 10 || col == 21, do_clip = col < 10;
    const int ch = is_y ? col - 22 : (col >= 11 ? col - 11 : col);
    const int stream = (blockIdx.x * 16 + wv) * 2 + half, nstream = gridDim.x * 32;
    double acc = 0.0;
    // a half-wave takes 32 CONSECUTIVE rows per step: row indices and weights arrive as one coalesced load each and are handed
    // round by shuffles, so the per-row loads (snow mean, this lane's element) of all 32 rows are independent and in flight together
    for (int base = stream * 32; base < nsample; base += nstream * 32) {
        const int mine = base + col;
        const int rr_l = mine < nsample ? rows[mine] : 0;
        const float w_l = mine < nsample ? weight[mine] : 0.f;
        const int cnt = min(32, nsample - base);
#pragma unroll 16
        for (int i = 0; i < 32; ++i) {
            const int rr = __shfl(rr_l, i, 32);
            const float w = __shfl(w_l, i, 32);
            const int dt = (rr >= npix) + (rr >= 2 * npix);          // a fit trains on at most 3 dates (CR.py:394-402)
            const int p = rr - dt * npix;
            const float sn = snowm[p];
            float z = is_y ? tiles[((long)(t0 + dt) * npix + p) * 10 + ch] : mosaic[(long)p * 10 + (is_sn ? 0 : ch)];
            if (do_clip) z = fminf(fmaxf(z, 0.005f), 1.0f);
            if (is_sn) z = sn;
            if (i < cnt) acc += (double)z * ((double)w * (double)sn);
        }
    }
    acc += __shfl_xor(acc, 32);
    if (lane < 32) red[wv][lane] = acc;
    __syncthreads();
    // (no grid-wide "last block" reduction here: its agent-scope release / acquire pair cost more than the products -- the
    // kernel stayed at 47-78 us whatever the main loop did; the consumer sums kSnowBlocks partials instead)
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        partial[blockIdx.x * 32 + threadIdx.x] = t;
    }
}
// Lawson-Hanson NNLS on the normal equations, ONE WAVE per band: lane r owns row r of the active system.  The
// arithmetic per matrix element is the same as a serial active-set solver's (row operations of the Gauss-Jordan elimination
// are independent per row); a single-lane version with fp64 arrays in scratch took 170-340 us per date, this one ~15 us.
__device__ __forceinline__ int wave_argmax_first(double v, bool eligible) {
    // lowest lane index among the eligible lanes holding the maximum; -1 if none
    double m = eligible ? v : -INFINITY;
    for (int k = 32; k >= 1; k >>= 1) m = fmax(m, __shfl_xor(m, k));
    const unsigned long long b = __ballot(eligible && v == m);
    return b ? __ffsll((long long)b) - 1 : -1;
}
__global__ void k_nnls(const double* __restrict__ Z, DatePlan* __restrict__ plan, Beta* __restrict__ be,
                       const double* __restrict__ snow_partial = nullptr, int snow_blocks = 0) {
    __shared__ double G[11][11], g[11], A[11][12], x[11], sf[11];
    __shared__ double sv[32];
    __shared__ int idx[11];
    const int band = blockIdx.x, lane = threadIdx.x;
    const int n = 11;
    if (!plan->proceed || plan->nrows <= 0) {          // nothing to fit for this date (CR.py:377-378)
        if (lane < n) be->b[band][lane] = 0.0;
        if (band == 0 && lane == 0) { be->fitted = 0; plan->fitted = 0; }
        return;
    }
    if (snow_partial) {                                  // date-batched form: Z holds everything but the snow products
        // per-block partials of k_gram_snow, summed in a fixed order (deterministic): two lanes per column, four independent
        // running sums each so that the loads overlap
        const int c = lane & 31, part = lane >> 5;
        double t = 0.0;
        for (int b0 = part; b0 < snow_blocks; b0 += 32) {            // 16 loads in flight, then a fixed-order sum
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (b0 + 2 * u < snow_blocks) ? snow_partial[(b0 + 2 * u) * 32 + c] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) t += v[u];
        }
        t += __shfl_xor(t, 32);
        if (lane < 32) sv[lane] = t;
        __syncthreads();
    }
    if (lane < n) {
        const int ci = (lane < band && lane < 10) ? lane : 11 + lane;            // CR.py:522 / :550
        const bool si = ci == 10 || ci == 21;
        for (int j = 0; j < n; ++j) {
            const int cj = (j < band && j < 10) ? j : 11 + j;
            const bool sj = cj == 10 || cj == 21;
            G[lane][j] = !snow_partial ? Z[ci * 32 + cj] : (si ? sv[sj ? 10 : cj] : (sj ? sv[ci] : Z[ci * 32 + cj]));
        }
        g[lane] = (snow_partial && si) ? sv[22 + band] : Z[ci * 32 + 22 + band];
        x[lane] = 0.0;
    }
    __syncthreads();
    unsigned P = 0;                                    // active set (uniform)
    const double tol = 1e-12 * fabs(g[0] + 1e-300) + 1e-15;
    for (int iter = 0; iter < 3 * n; ++iter) {
        double w = 0.0;
        if (lane < n) { w = g[lane]; for (int j = 0; j < n; ++j) w -= G[lane][j] * x[j]; }
        const int best = wave_argmax_first(w, lane < n && !((P >> lane) & 1u) && w > tol);
        if (best < 0) break;
        P |= 1u << best;
        for (int inner = 0; inner < 3 * n; ++inner) {
            const int m = __popc(P);
            if (lane < n && ((P >> lane) & 1u)) idx[__popc(P & ((1u << lane) - 1u))] = lane;
            __syncthreads();
            if (lane < m) {
                for (int c2 = 0; c2 < m; ++c2) A[lane][c2] = G[idx[lane]][idx[c2]];
                A[lane][m] = g[idx[lane]];
            }
            __syncthreads();
            for (int c2 = 0; c2 < m; ++c2) {
                // partial pivoting: first row r >= c2 with the largest |A[r][c2]|
                const int piv = wave_argmax_first(lane < m ? fabs(A[lane][c2]) : 0.0, lane >= c2 && lane < m);
                if (piv != c2 && lane <= m) { const double tv = A[piv][lane]; A[piv][lane] = A[c2][lane]; A[c2][lane] = tv; }
                __syncthreads();
                const double d = A[c2][c2];
                if (fabs(d) >= 1e-300 && lane < m && lane != c2) {
                    const double f = A[lane][c2] / d;
                    if (f != 0.0) for (int k = c2; k <= m; ++k) A[lane][k] -= f * A[c2][k];
                }
                __syncthreads();
            }
            double sv = 0.0;
            if (lane < m) { const double d = A[lane][lane]; sv = fabs(d) < 1e-300 ? 0.0 : A[lane][m] / d; }
            if (lane < n) sf[lane] = 0.0;
            __syncthreads();
            if (lane < m) sf[idx[lane]] = sv;
            __syncthreads();
            const bool allpos = __ballot(lane < m && sv <= 0.0) == 0ull;
            if (allpos) {
                if (lane < n) x[lane] = ((P >> lane) & 1u) ? sf[lane] : 0.0;
                __syncthreads();
                break;
            }
            double a = 1.0;
            if (lane < n && ((P >> lane) & 1u) && sf[lane] <= 0.0) a = x[lane] / (x[lane] - sf[lane]);
            for (int k = 32; k >= 1; k >>= 1) a = fmin(a, __shfl_xor(a, k));
            bool drop = false;
            if (lane < n && ((P >> lane) & 1u)) {
                const double xn = x[lane] + a * (sf[lane] - x[lane]);
                drop = xn <= 1e-15;
                x[lane] = drop ? 0.0 : xn;
            }
            P &= ~(unsigned)__ballot(drop);
            __syncthreads();
        }
    }
    if (lane < n) be->b[band][lane] = x[lane];
    if (band == 0 && lane == 0) { be->fitted = 1; plan->fitted = 1; }
}
__global__ void k_cloud_thresholds(const SelState* __restrict__ st, const int* __restrict__ n_only, int npix, float* __restrict__ thr) {
    if (threadIdx.x) return;
    const long long n = (long long)npix - *n_only;
    if (n <= 0) { thr[0] = INFINITY; thr[1] = INFINITY; return; }                 // CR.py:718-719: no reference pixels
    const double pos = 0.99 * (double)(n - 1), fr = pos - floor(pos);
    for (int k = 0; k < 2; ++k) {
        const double a = fkey_inv(st[2 * k].prefix), b = fkey_inv(st[2 * k + 1].prefix);
        thr[k] = (float)(a + (b - a) * fr);
    }
}

}  // namespace

#define GF_T(kern, T, ...)                                                            \
    do {                                                                              \
        if ((T) <= 8) hipLaunchKernelGGL((kern<8>), __VA_ARGS__);                     \
        else if ((T) <= 16) hipLaunchKernelGGL((kern<16>), __VA_ARGS__);              \
        else hipLaunchKernelGGL((kern<32>), __VA_ARGS__);                             \
    } while (0)
#define GF_T2(kern, flag, T, ...)                                                     \
    do {                                                                              \
        if ((T) <= 8) hipLaunchKernelGGL((kern<8, flag>), __VA_ARGS__);               \
        else if ((T) <= 16) hipLaunchKernelGGL((kern<16, flag>), __VA_ARGS__);        \
        else hipLaunchKernelGGL((kern<32, flag>), __VA_ARGS__);                       \
    } while (0)

// diamond dilation of one [X, Y] byte plane, radius r <= kDilR (two separable passes; in, out and the scratch plane are distinct)
static ttc_status dilate_diamond(ttc_ctx* c, const unsigned char* in, int X, int Y, int r, int invert, unsigned char* out, hipStream_t s) {
    if (r > kDilR) {                       // not used by the path: the direct form handles any radius
        hipLaunchKernelGGL(k_dilate_diamond, dim3((unsigned)(((long)X * Y + 255) / 256)), dim3(256), 0, s, in, X, Y, r, invert, out);
        return TTC_OK;
    }
    unsigned char* g = static_cast<unsigned char*>(c->scratch_buf("gf_dil_rows", (size_t)X * Y));
    if (!g) return c->fail(TTC_ERR_NOMEM, "dilation scratch");
    hipLaunchKernelGGL(k_dil_rows4, dim3((unsigned)(((long)X * ((Y + 3) / 4) + 255) / 256)), dim3(256), 0, s, in, X, Y, r, invert, g);
    hipLaunchKernelGGL(k_dil_cols4, dim3((unsigned)(((long)((X + 3) / 4) * Y + 255) / 256)), dim3(256), 0, s, g, X, Y, r, out);
    return TTC_OK;
}

// a6 ------------------------------------------------------------------------------------------------
ttc_status gapfill_feather(ttc_ctx* c, const float* d_mask, int T, int X, int Y, int closing, int clip, float* d_w,
                           hipStream_t s) {
    if (!d_mask || !d_w || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "feather: bad argument (T in [1,32])");
    if (closing != 15 && closing != 20) return c->fail(TTC_ERR_ARG, "feather: closing must be 15 or 20");
    const long npix = (long)X * Y;
    unsigned char* b0 = static_cast<unsigned char*>(c->scratch_buf("gf_g2", 2 * (size_t)T * npix));      // two byte planes per date
    unsigned char* b1 = b0 ? b0 + (size_t)T * npix : nullptr;
    int* nz = static_cast<int*>(c->scratch_buf("gf_nz", sizeof(int) * kMaxT));
    if (!b0 || !nz) return c->fail(TTC_ERR_NOMEM, "feather scratch");
    KTimer kt(c, "feather", s);
    const dim3 grid((unsigned)((npix + 255) / 256), T), blk(256);
    TTC_HIP(c, hipMemsetAsync(nz, 0, sizeof(int) * kMaxT, s));
    hipLaunchKernelGGL(k_mask_positive, dim3(96, T), blk, 0, s, d_mask, (int)npix, clip, nz);
    hipLaunchKernelGGL(k_edt_rows4, dim3((unsigned)(((long)X * ((Y + 3) / 4) + 255) / 256), T), blk, 0, s, d_mask, X, Y, clip, b1);
    hipLaunchKernelGGL(k_edt_cols4, dim3((unsigned)(((long)((X + 3) / 4) * Y + 255) / 256), T), blk, 0, s, b1, X, Y, b0);
    // scipy grey_closing(size): dilation window [-(size/2 - 1), size/2] for even sizes, then erosion [-size/2, size/2 - 1]
    const int dlo = closing == 20 ? -9 : -7, dhi = closing == 20 ? 10 : 7, elo = closing == 20 ? -10 : -7, ehi = closing == 20 ? 9 : 7;
    if (X >= 32 && Y >= 32) {            // four outputs per thread (the reflect index below is single-fold: needs the plane larger than the window)
        const dim3 gy((unsigned)(((long)X * ((Y + 3) / 4) + 255) / 256), T), gx((unsigned)(((long)((X + 3) / 4) * Y + 255) / 256), T);
        if (closing == 20) {
            hipLaunchKernelGGL((k_minmax4<true, true, 20>), gy, blk, 0, s, b0, X, Y, dlo, b1);
            hipLaunchKernelGGL((k_minmax4<true, false, 20>), gx, blk, 0, s, b1, X, Y, dlo, b0);
            hipLaunchKernelGGL((k_minmax4<false, true, 20>), gy, blk, 0, s, b0, X, Y, elo, b1);
            hipLaunchKernelGGL((k_minmax4<false, false, 20>), gx, blk, 0, s, b1, X, Y, elo, b0);
        } else {
            hipLaunchKernelGGL((k_minmax4<true, true, 15>), gy, blk, 0, s, b0, X, Y, dlo, b1);
            hipLaunchKernelGGL((k_minmax4<true, false, 15>), gx, blk, 0, s, b1, X, Y, dlo, b0);
            hipLaunchKernelGGL((k_minmax4<false, true, 15>), gy, blk, 0, s, b0, X, Y, elo, b1);
            hipLaunchKernelGGL((k_minmax4<false, false, 15>), gx, blk, 0, s, b1, X, Y, elo, b0);
        }
    } else {
    hipLaunchKernelGGL((k_minmax<true, true>), grid, blk, 0, s, b0, X, Y, dlo, dhi, b1);
    hipLaunchKernelGGL((k_minmax<true, false>), grid, blk, 0, s, b1, X, Y, dlo, dhi, b0);
    hipLaunchKernelGGL((k_minmax<false, true>), grid, blk, 0, s, b0, X, Y, elo, ehi, b1);
    hipLaunchKernelGGL((k_minmax<false, false>), grid, blk, 0, s, b1, X, Y, elo, ehi, b0);
    }
    hipLaunchKernelGGL(k_feather_store, grid, blk, 0, s, b0, d_mask, nz, (int)npix, clip, d_w);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// a7 ------------------------------------------------------------------------------------------------
static ttc_status water_mask(ttc_ctx* c, const float* d_tiles, int T, int X, int Y, bool of_median, bool dilate,
                             unsigned char* out, hipStream_t s) {
    const int npix = X * Y;
    unsigned char* tmp = static_cast<unsigned char*>(c->scratch_buf("gf_wtmp", 2 * (size_t)npix));
    if (!tmp) return c->fail(TTC_ERR_NOMEM, "water scratch");
    const dim3 grid((npix + 255) / 256), blk(256);
    unsigned char* raw = dilate ? tmp : out;
    if (of_median) GF_T2(k_water, true, T, grid, blk, 0, s, d_tiles, T, npix, raw);
    else GF_T2(k_water, false, T, grid, blk, 0, s, d_tiles, T, npix, raw);
    if (dilate) {   // binary_dilation(1 - water, 2) then binary_dilation(1 - that, 5)  (CR.py:585-586)
        TTC_CHECK(dilate_diamond(c, raw, X, Y, 2, 1, tmp + npix, s));
        TTC_CHECK(dilate_diamond(c, tmp + npix, X, Y, 5, 1, out, s));
    }
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// Date-by-date form: exactly the reference's loop, including the `interp[i] = 1` side effect of a date that cannot be
// aligned (CR.py:679-680), which changes the reference set of every LATER date.  Used when the batched form below
// finds such a date.
static ttc_status aligned_mosaic_sequential(ttc_ctx* c, const float* d_tiles, float* d_w, int T, int X, int Y, float* d_mosaic,
                                            hipStream_t s) {
    const int npix = X * Y;
    unsigned char* water = static_cast<unsigned char*>(c->scratch_buf("gf_water", (size_t)npix));
    unsigned char* valid = static_cast<unsigned char*>(c->scratch_buf("gf_valid", (size_t)npix));
    float* ref = static_cast<float*>(c->scratch_buf("gf_ref", sizeof(float) * 10 * (size_t)npix));
    float* divisor = static_cast<float*>(c->scratch_buf("gf_div", sizeof(float) * (size_t)npix));
    char* small = static_cast<char*>(c->scratch_buf("gf_small", 65536));
    if (!water || !valid || !ref || !divisor || !small) return c->fail(TTC_ERR_NOMEM, "aligned_mosaic scratch");
    SelState* st = reinterpret_cast<SelState*>(small);                   // 40 problems
    unsigned* hist = reinterpret_cast<unsigned*>(small + 4096);           // 40 * 256 * 4 = 40960
    double* mean = reinterpret_cast<double*>(small + 4096 + 40960);       // 20
    double* var = mean + 20;                                              // 20
    int* count = reinterpret_cast<int*>(var + 20);                        // count, n_land
    AlignPar* ap = reinterpret_cast<AlignPar*>(count + 4);
    TTC_CHECK(water_mask(c, d_tiles, T, X, Y, false, true, water, s));
    const dim3 grid((npix + 255) / 256), blk(256);
    TTC_HIP(c, hipMemsetAsync(d_mosaic, 0, sizeof(float) * 10 * (size_t)npix, s));
    TTC_HIP(c, hipMemsetAsync(count, 0, sizeof(int) * 4, s));
    hipLaunchKernelGGL(k_divisor, grid, blk, 0, s, d_w, T, npix, divisor);
    hipLaunchKernelGGL(k_count_land, dim3(64), blk, 0, s, water, npix, count + 1);
    TTC_HIP(c, hipMemsetAsync(hist, 0, 40960, s));
    for (int i = 0; i < T; ++i) {
        TTC_HIP(c, hipMemsetAsync(count, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_mosaic_ref, grid, blk, 0, s, d_tiles, d_w, water, T, npix, i, ref, valid, count);
        // device-side control: selection ranks, means and the n > 1000 decision all read the device count,
        // so the whole mosaic is enqueued without a host round trip
        hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(64), 0, s, st, 40, count, 0, 0, 0, PctList{});
        SrcMosaic src{ref, d_tiles + (long)i * npix * 10, valid, npix};
        TTC_HIP(c, radix_select(src, st, hist, 40, s));
        TTC_HIP(c, hipMemsetAsync(mean, 0, sizeof(double) * 40, s));
        hipLaunchKernelGGL(k_stat_sum, dim3(96, 20), dim3(256), 0, s, src, (const double*)nullptr, count, mean);
        hipLaunchKernelGGL(k_stat_sum, dim3(96, 20), dim3(256), 0, s, src, (const double*)mean, count, var);
        hipLaunchKernelGGL(k_align_params, dim3(1), dim3(64), 0, s, st, var, count, count + 1, ap);
        hipLaunchKernelGGL(k_mosaic_accum, grid, blk, 0, s, d_tiles, d_w, water, ap, npix, i, d_mosaic);
        TTC_HIP(c, hipGetLastError());
    }
    GF_T(k_mosaic_final, T, grid, blk, 0, s, d_tiles, divisor, T, npix, d_mosaic);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// ---- batched form: all T dates at once ---------------------------------------------------------------------------
// Without an unalignable date the T iterations of make_aligned_mosaic are independent, so every stage runs once with
// the date as a grid dimension: ~20 launches instead of ~20 per date, one pass over the stack for the T reference
// means (the per-date form re-reads all T dates T times), single-pass statistics, and 20 selection problems per date
// (the upper median comes from one extra counting pass instead of a second radix select).
template <int TM>
__global__ void k_ref_all(const float* __restrict__ tiles, const float* __restrict__ w, const unsigned char* __restrict__ water,
                          int T, int npix, float* __restrict__ ref_all, unsigned* __restrict__ vmask, int* __restrict__ count) {
#pragma clang fp contract(off)
    // thread = the float2 of two neighbouring bands of one pixel (round 5; one float per thread before: twice the load / store instructions)
    __shared__ int cnt[TM];
    if (threadIdx.x < TM) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long total2 = (long)npix * 5;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id - threadIdx.x < total2; id += (long)gridDim.x * blockDim.x) {
        unsigned mask = 0;
        const bool live = id < total2;
        const int p = live ? (int)(id / 5) : 0;
        const int c2 = (int)(id - (long)p * 5);
        if (live) {
            float wv[TM];
            float2 v[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                wv[t] = t < T ? w[(long)t * npix + p] : 1.0f;
                v[t] = t < T ? reinterpret_cast<const float2*>(tiles + (long)t * npix * 10)[id] : float2{0.0f, 0.0f};
            }
            if (!water[p]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (i < T && wv[i] < 0.25f) {
                        float2 sum = {0.f, 0.f};
                        int n = 0;
#pragma unroll
                        for (int b = 0; b < TM; ++b)
                            if (b < T && b != i && wv[b] < 1.0f) { sum.x += v[b].x; sum.y += v[b].y; ++n; }
                        if (n > 0) { mask |= 1u << i; reinterpret_cast<float2*>(ref_all + (long)i * npix * 10)[id] = float2{sum.x / (float)n, sum.y / (float)n}; }
                    }
                }
            }
            if (c2 == 0) vmask[p] = mask;
        }
        for (int i = 0; i < T; ++i) {
            const int k = __popcll(__ballot(live && c2 == 0 && ((mask >> i) & 1u)));
            if ((threadIdx.x & 63) == 0 && k) atomicAdd(&cnt[i], k);
        }
    }
    __syncthreads();
    if (threadIdx.x < T && cnt[threadIdx.x]) atomicAdd(&count[threadIdx.x], cnt[threadIdx.x]);
}
// problems q = date*20 + band*2 + which (0 = reference mean, 1 = the date itself); rank = lower median
__global__ void k_sel_init_dates(SelState* __restrict__ st, const int* __restrict__ count, int T) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= T * 20) return;
    const int n = count[q / 20];
    SelState ss; ss.prefix = 0; ss.mask = 0; ss.k = n > 0 ? (n - 1) / 2 : 0;
    st[q] = ss;
}
// per (date, column): sum, sum of squares (double) over the valid rows; the key of the UPPER median (order statistic n / 2)
struct ColStat { double sum, sq; unsigned min_above, up_key; };
__global__ void k_stat_init(ColStat* cs) { cs[threadIdx.x].sum = 0.0; cs[threadIdx.x].sq = 0.0; cs[threadIdx.x].min_above = 0xffffffffu; cs[threadIdx.x].up_key = 0; }
// One radix-select pass for the 20 problems (10 bands x {reference, date}) of date blockIdx.y.  Histograms live in LDS;
// a lane aggregates its OWN consecutive hits of one bin in registers and touches LDS only when the bin changes: on the high
// passes reflectances share their top key bytes from pixel to pixel (a handful of bins), on the low passes few keys still match
// the prefix -- either way almost no atomics, and no wave collectives.
// Round 4: the separate statistics pass over the stack (k_stat_all, 0.2 ms per tile: sums, sums of squares, and the count / successor
// search that turned the lower median into the upper one) is gone.  MODE 1 (the first pass, shift 24) also accumulates sum and sum
// of squares of every column in double -- they do not depend on the median; MODE 2 (the last pass, shift 0) also keeps, per column,
// the smallest key ABOVE the 24-bit prefix range: with the last byte's histogram that is all k_pick_last needs to name the order
// statistic that follows the lower median.
template <int MODE>
__global__ __launch_bounds__(256) void k_hist_all(const float* __restrict__ ref_all, const float* __restrict__ tiles,
                                                   const unsigned* __restrict__ vmask, int npix, const SelState* __restrict__ st,
                                                   int shift, unsigned* __restrict__ hist, ColStat* __restrict__ cs) {
    __shared__ unsigned h[20 * 256];
    __shared__ unsigned pf[20], mk[20];
    __shared__ double ssum[20], ssq[20];
    __shared__ unsigned smin[20];
    const int i = blockIdx.y;
    for (int k = threadIdx.x; k < 20 * 256; k += blockDim.x) h[k] = 0;
    if (threadIdx.x < 20) {
        pf[threadIdx.x] = st[i * 20 + threadIdx.x].prefix; mk[threadIdx.x] = st[i * 20 + threadIdx.x].mask;
        ssum[threadIdx.x] = 0.0; ssq[threadIdx.x] = 0.0; smin[threadIdx.x] = 0xffffffffu;
    }
    __syncthreads();
    int cur[20];
    unsigned cnt[20];
    double a1[MODE == 1 ? 20 : 1], a2[MODE == 1 ? 20 : 1];
    unsigned mn[MODE == 2 ? 20 : 1];
#pragma unroll
    for (int q = 0; q < 20; ++q) { cur[q] = -1; cnt[q] = 0; }
#pragma unroll
    for (int q = 0; q < (MODE == 1 ? 20 : 1); ++q) { a1[q] = 0.0; a2[q] = 0.0; }
#pragma unroll
    for (int q = 0; q < (MODE == 2 ? 20 : 1); ++q) mn[q] = 0xffffffffu;
    const int stride = gridDim.x * blockDim.x;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        if (!((vmask[p] >> i) & 1u)) continue;
        const float2* r = reinterpret_cast<const float2*>(ref_all + ((long)i * npix + p) * 10);
        const float2* sv = reinterpret_cast<const float2*>(tiles + ((long)i * npix + p) * 10);
#pragma unroll
        for (int c2 = 0; c2 < 5; ++c2) {
            const float2 a = r[c2], b = sv[c2];
            const float v[4] = {a.x, b.x, a.y, b.y};         // problems 4*c2 .. 4*c2 + 3 = (ch, ref), (ch, date), (ch+1, ref), (ch+1, date)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * c2 + j;
                if (MODE == 1) { a1[q] += (double)v[j]; a2[q] += (double)v[j] * (double)v[j]; }
                const unsigned k = fkey(v[j]);
                if ((k & mk[q]) != pf[q]) {
                    if (MODE == 2 && k > (pf[q] | 255u)) mn[q] = min(mn[q], k);
                    continue;
                }
                const int bin = (int)((k >> shift) & 255u);
                if (bin == cur[q]) { cnt[q]++; continue; }
                if (cnt[q]) atomicAdd(&h[q * 256 + cur[q]], cnt[q]);
                cur[q] = bin; cnt[q] = 1;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 20; ++q)
        if (cnt[q]) atomicAdd(&h[q * 256 + cur[q]], cnt[q]);
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            for (int k = 32; k >= 1; k >>= 1) { a1[q] += __shfl_xor(a1[q], k); a2[q] += __shfl_xor(a2[q], k); }
            if ((threadIdx.x & 63) == 0) { atomicAdd(&ssum[q], a1[q]); atomicAdd(&ssq[q], a2[q]); }
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            for (int k = 32; k >= 1; k >>= 1) mn[q] = min(mn[q], (unsigned)__shfl_xor((int)mn[q], k));
            if ((threadIdx.x & 63) == 0 && mn[q] != 0xffffffffu) atomicMin(&smin[q], mn[q]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 20 * 256; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[(long)i * 20 * 256 + k], h[k]);
    if (MODE == 1 && threadIdx.x < 20) { atomicAdd(&cs[i * 20 + threadIdx.x].sum, ssum[threadIdx.x]); atomicAdd(&cs[i * 20 + threadIdx.x].sq, ssq[threadIdx.x]); }
    if (MODE == 2 && threadIdx.x < 20 && smin[threadIdx.x] != 0xffffffffu) atomicMin(&cs[i * 20 + threadIdx.x].min_above, smin[threadIdx.x]);
}
// The last pick (shift 0) of the date problems: the lower median like k_sel_pick, plus the key of the order statistic that follows it
// (the upper median of an even count): the same key when its bin holds another element, else the next non-empty bin of this last-byte
// histogram, else the smallest key above the whole 24-bit prefix range (ColStat.min_above).
__global__ void k_pick_last(SelState* __restrict__ st, unsigned* __restrict__ hist, ColStat* __restrict__ cs) {
    const int q = blockIdx.x, lane = threadIdx.x;
    unsigned c[4];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = hist[q * 256 + 4 * lane + j]; mine += c[j]; hist[q * 256 + 4 * lane + j] = 0; }
    unsigned incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const long long excl = (long long)incl - mine;
    SelState ss = st[q];
    const long long k = ss.k;
    const bool here = k >= excl && k < (long long)incl;
    const unsigned long long m = __ballot(here);
    const int owner = m ? __ffsll((long long)m) - 1 : 63;
    // per lane: the first non-empty bin of this lane (for the successor search)
    int first_bin = -1;
#pragma unroll
    for (int j = 3; j >= 0; --j) if (c[j]) first_bin = 4 * lane + j;
    int b = 0;
    long long r = 0;
    if (lane == owner) {
        r = k - excl;
        for (; b < 3; ++b) { if (r < (long long)c[b]) break; r -= c[b]; }
    }
    b = __shfl(b, owner);
    const long long rr = __shfl((int)r, owner);
    const unsigned cb = __shfl((int)c[0], owner) * (b == 0) + __shfl((int)c[1], owner) * (b == 1) + __shfl((int)c[2], owner) * (b == 2) + __shfl((int)c[3], owner) * (b == 3);
    const int lo_bin = 4 * owner + b;
    // next non-empty bin above lo_bin: inside the owner lane, else the first bin of the first later lane that has any
    int nxt = -1;
    if (lane == owner) { for (int j = 3; j > b; --j) if (c[j]) nxt = 4 * lane + j; }
    nxt = __shfl(nxt, owner);
    const unsigned long long later = __ballot(lane > owner && first_bin >= 0);
    if (nxt < 0 && later) nxt = __shfl(first_bin, __ffsll((long long)later) - 1);
    if (lane == 0) {
        const unsigned lo_key = ss.prefix | (unsigned)lo_bin;
        unsigned up_key = lo_key;
        if (!(m && rr + 1 < (long long)cb)) up_key = nxt >= 0 ? (ss.prefix | (unsigned)nxt) : cs[q].min_above;
        ss.prefix = lo_key; ss.mask |= 255u; ss.k = m ? rr : 0;
        st[q] = ss;
        cs[q].up_key = up_key;
    }
}
// ---- medians from a SAMPLED BRACKET (round 5): one pass over the stack instead of four ---------------------------------------------
// The four radix-select passes above read the 2 x 183 MB of (reference, date) columns four times for 240 medians.  Here: (1) a systematic
// sample of kMedS pixels per date gives, per problem, two keys [lo, hi] that bracket the median's rank with 5 sigma of the sample quantile's
// spread (k_med_sample / k_med_bracket: ~4 % of the rows fall inside); (2) ONE pass over the stack counts the keys below lo, collects the
// keys inside the bracket (staged in LDS per workgroup, appended with one global atomic per workgroup and problem) and accumulates the
// column moments exactly like k_hist_all<1>; (3) one workgroup per problem selects the order statistics k and k + 1 among the candidates in
// registers (k_med_final).  The result is the EXACT order statistic -- the same key the radix select finds -- because the rank bookkeeping
// (below + position among the candidates) is exact; if the bracket misses (or a staging area overflows) that problem's workgroup falls
// back to a radix select over the full column by itself.  TTC_MEDIAN_RADIX=1 runs the four-pass form, TTC_MEDIAN_FORCE_FALLBACK=1 empties
// every bracket (tests: all three give identical keys).
constexpr int kMedS = 16384;        // sample pixels per date
constexpr int kMedCap = 24576;      // candidate keys per problem (k_med_final holds them in registers: 96 per thread)
constexpr int kMedStage = 512;      // candidates one workgroup of the counting pass stages per problem
struct MedBracket { unsigned lo, hi; };                 // lo > hi: no bracket
struct MedCount { unsigned below, ncand, fail, fell_back; };

__global__ __launch_bounds__(256) void k_med_sample(const float* __restrict__ ref_all, const float* __restrict__ tiles,
                                                     const unsigned* __restrict__ vmask, int npix, unsigned* __restrict__ samp,
                                                     int* __restrict__ nsamp) {
    __shared__ int spix[256];
    __shared__ int sbase, scount;
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) scount = 0;
    __syncthreads();
    const int p = (int)((long long)j * npix / kMedS);
    const bool valid = j < kMedS && ((vmask[p] >> i) & 1u);
    const unsigned long long b = __ballot(valid);
    int base = 0;
    if (b && lane == __ffsll((long long)b) - 1) base = atomicAdd(&scount, __popcll(b));      // order inside the sample does not matter
    base = __shfl(base, b ? __ffsll((long long)b) - 1 : 0);
    if (valid) spix[base + __popcll(b & ((1ull << lane) - 1ull))] = p;
    __syncthreads();
    const int n = scount;
    if (threadIdx.x == 0 && n) sbase = atomicAdd(&nsamp[i], n);
    __syncthreads();
    // five lanes per sampled pixel: lane c2 reads the float2 of bands (2 c2, 2 c2 + 1) of the reference and of the date (40 contiguous
    // bytes per pixel and array), and writes four keys
    for (int w = threadIdx.x; w < n * 5; w += blockDim.x) {
        const int e = w / 5, c2 = w - e * 5;
        const int slot = sbase + e, pp = spix[e];
        const float2 r = reinterpret_cast<const float2*>(ref_all + ((long)i * npix + pp) * 10)[c2];
        const float2 d = reinterpret_cast<const float2*>(tiles + ((long)i * npix + pp) * 10)[c2];
        unsigned* o = samp + ((long)(i * 20 + 4 * c2)) * kMedS + slot;
        o[0] = fkey(r.x); o[kMedS] = fkey(d.x); o[2 * kMedS] = fkey(r.y); o[3 * kMedS] = fkey(d.y);
    }
}
// rank-th smallest (0-based) of the keys get(j), j < m, by a 256-thread workgroup: four 8-bit passes with an LDS histogram; h = 260 words
template <class GET>
__device__ unsigned block_select(GET get, int m, long long rank, unsigned* h) {
    unsigned prefix = 0, mask = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int shift = 24; shift >= 0; shift -= 8) {
        h[tid] = 0;
        __syncthreads();
        // a lane aggregates its own consecutive hits of one bin (k_hist_all's device): keys of one column share their high bytes, so on the
        // high passes every hit lands in a handful of bins (one LDS address per wave instruction otherwise), on the low ones few keys match
        int cur = -1;
        unsigned cnt = 0;
        for (int j = tid; j < m; j += 256) {
            unsigned k;
            if (!get(j, k) || (k & mask) != prefix) continue;
            const int bin = (int)((k >> shift) & 255u);
            if (bin == cur) { ++cnt; continue; }
            if (cnt) atomicAdd(&h[cur], cnt);
            cur = bin; cnt = 1;
        }
        if (cnt) atomicAdd(&h[cur], cnt);
        __syncthreads();
        if (tid < 64) {                                  // lane l owns bins 4l .. 4l + 3 (k_sel_pick's scan)
            unsigned c[4], mine = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = h[4 * lane + q]; mine += c[q]; }
            unsigned incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const long long excl = (long long)incl - mine;
            const bool here = rank >= excl && rank < (long long)incl;
            const unsigned long long mm = __ballot(here);
            const int owner = mm ? __ffsll((long long)mm) - 1 : 63;
            if (lane == owner) {
                long long r = rank - excl;
                int b = 0;
                for (; b < 3; ++b) { if (r < (long long)c[b]) break; r -= c[b]; }
                h[256] = (unsigned)(4 * lane + b);
                h[257] = (unsigned)(mm ? r : 0);
            }
        }
        __syncthreads();
        prefix |= h[256] << shift; mask |= 255u << shift; rank = (long long)h[257];
        __syncthreads();
    }
    return prefix;
}
// the same select over keys the workgroup holds in REGISTERS (thread t owns keys t, t + 256, ...: NPER per thread, m in total): the eight
// passes of two selects then touch no memory but the LDS histogram (from L2 / LDS each pass re-waited the load latency: 83-96 us per launch)
template <int NPER>
__device__ unsigned block_select_reg(const unsigned (&key)[NPER], int m, long long rank, unsigned* h) {
    unsigned prefix = 0, mask = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int shift = 24; shift >= 0; shift -= 8) {
        h[tid] = 0;
        __syncthreads();
        int cur = -1;
        unsigned cnt = 0;
#pragma unroll
        for (int u = 0; u < NPER; ++u) {
            const unsigned k = key[u];
            if (u * 256 + tid >= m || (k & mask) != prefix) continue;
            const int bin = (int)((k >> shift) & 255u);
            if (bin == cur) { ++cnt; continue; }
            if (cnt) atomicAdd(&h[cur], cnt);
            cur = bin; cnt = 1;
        }
        if (cnt) atomicAdd(&h[cur], cnt);
        __syncthreads();
        if (tid < 64) {
            unsigned c[4], mine = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = h[4 * lane + q]; mine += c[q]; }
            unsigned incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const long long excl = (long long)incl - mine;
            const bool here = rank >= excl && rank < (long long)incl;
            const unsigned long long mm = __ballot(here);
            const int owner = mm ? __ffsll((long long)mm) - 1 : 63;
            if (lane == owner) {
                long long r = rank - excl;
                int b = 0;
                for (; b < 3; ++b) { if (r < (long long)c[b]) break; r -= c[b]; }
                h[256] = (unsigned)(4 * lane + b);
                h[257] = (unsigned)(mm ? r : 0);
            }
        }
        __syncthreads();
        prefix |= h[256] << shift; mask |= 255u << shift; rank = (long long)h[257];
        __syncthreads();
    }
    return prefix;
}
__global__ __launch_bounds__(256) void k_med_bracket(const unsigned* __restrict__ samp, const int* __restrict__ nsamp,
                                                      const int* __restrict__ count, MedBracket* __restrict__ br, int force_fail) {
    __shared__ unsigned h[260];
    const int q = blockIdx.x, i = q / 20;
    const int m = min(nsamp[i], kMedS), n = count[i];
    MedBracket out{0u, 0xffffffffu};                        // no usable sample: every key is a candidate (fits when n is small)
    if (force_fail) { out.lo = 1u; out.hi = 0u; }
    else if (m > 0 && n > 0) {
        const unsigned* keys = samp + (long)q * kMedS;
        unsigned key[kMedS / 256];
#pragma unroll
        for (int u = 0; u < kMedS / 256; ++u) key[u] = (u * 256 + (int)threadIdx.x < m) ? keys[u * 256 + threadIdx.x] : 0u;
        const double k0 = (double)((n - 1) / 2);
        const double pos = (k0 + 0.5) * (double)m / (double)n, delta = 2.5 * sqrt((double)m) + 2.0;
        const long long rlo = (long long)floor(pos - delta), rhi = (long long)ceil(pos + delta) + 1;
        if (rlo >= 0) out.lo = block_select_reg(key, m, rlo, h);
        if (rhi < m) out.hi = block_select_reg(key, m, rhi, h);
    }
    if (threadIdx.x == 0) br[q] = out;
}
__global__ __launch_bounds__(256) void k_med_count(const float* __restrict__ ref_all, const float* __restrict__ tiles,
                                                    const unsigned* __restrict__ vmask, int npix, const MedBracket* __restrict__ br,
                                                    MedCount* __restrict__ mc, unsigned* __restrict__ cand, ColStat* __restrict__ cs) {
    __shared__ unsigned stage[20 * kMedStage];
    __shared__ unsigned lo[20], hi[20], scnt[20], sbelow[20], sbase[20];
    __shared__ double ssum[20], ssq[20];
    const int i = blockIdx.y;
    if (threadIdx.x < 20) {
        const MedBracket b = br[i * 20 + threadIdx.x];
        lo[threadIdx.x] = b.lo; hi[threadIdx.x] = b.hi; scnt[threadIdx.x] = 0; sbelow[threadIdx.x] = 0;
        ssum[threadIdx.x] = 0.0; ssq[threadIdx.x] = 0.0;
    }
    __syncthreads();
    unsigned below[20];
    double a1[20], a2[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) { below[q] = 0; a1[q] = 0.0; a2[q] = 0.0; }
    // (an LDS-staged loader -- the workgroup copies its 256 consecutive 40-byte records with coalesced loads, lanes read them from LDS --
    // measured SLOWER here: 282 vs 196 us; 60 KB of LDS per workgroup leaves two workgroups per CU and nothing to hide the copy's latency)
    // the 20 brackets are workgroup-uniform: read through a uniform pointer they live in SGPRs (from LDS: 40 ds_reads per pixel)
    MedBracket bq[20];
#pragma unroll
    for (int q = 0; q < 20; ++q) bq[q] = br[i * 20 + q];
    const int stride = gridDim.x * blockDim.x;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        if (!((vmask[p] >> i) & 1u)) continue;
        const float2* r = reinterpret_cast<const float2*>(ref_all + ((long)i * npix + p) * 10);
        const float2* sv = reinterpret_cast<const float2*>(tiles + ((long)i * npix + p) * 10);
#pragma unroll
        for (int c2 = 0; c2 < 5; ++c2) {
            const float2 a = r[c2], b = sv[c2];
            const float v[4] = {a.x, b.x, a.y, b.y};         // problems 4*c2 .. 4*c2 + 3, as in k_hist_all
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * c2 + j;
                a1[q] += (double)v[j]; a2[q] += (double)v[j] * (double)v[j];
                const unsigned k = fkey(v[j]);
                if (k < bq[q].lo) { below[q]++; continue; }
                if (k > bq[q].hi) continue;
                const unsigned slot = atomicAdd(&scnt[q], 1u);
                if (slot < (unsigned)kMedStage) stage[q * kMedStage + slot] = k;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 20; ++q) {
        for (int k = 32; k >= 1; k >>= 1) {
            a1[q] += __shfl_xor(a1[q], k); a2[q] += __shfl_xor(a2[q], k);
            below[q] += (unsigned)__shfl_xor((int)below[q], k);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&ssum[q], a1[q]); atomicAdd(&ssq[q], a2[q]); if (below[q]) atomicAdd(&sbelow[q], below[q]); }
    }
    __syncthreads();
    if (threadIdx.x < 20) {
        const int q = threadIdx.x;
        MedCount* m = mc + i * 20 + q;
        atomicAdd(&cs[i * 20 + q].sum, ssum[q]); atomicAdd(&cs[i * 20 + q].sq, ssq[q]);
        if (sbelow[q]) atomicAdd(&m->below, sbelow[q]);
        unsigned c = scnt[q];
        if (c > (unsigned)kMedStage) { atomicOr(&m->fail, 1u); c = kMedStage; }
        sbase[q] = c ? atomicAdd(&m->ncand, c) : 0u;
        if (c && sbase[q] + c > (unsigned)kMedCap) atomicOr(&m->fail, 1u);
        scnt[q] = c;
    }
    __syncthreads();
    for (int q = 0; q < 20; ++q) {
        const unsigned c = scnt[q], base = sbase[q];
        unsigned* dst = cand + (long)(i * 20 + q) * kMedCap;
        for (unsigned j = threadIdx.x; j < c; j += blockDim.x)
            if (base + j < (unsigned)kMedCap) dst[base + j] = stage[q * kMedStage + j];
    }
}
__global__ __launch_bounds__(256) void k_med_final(const unsigned* __restrict__ cand, MedCount* __restrict__ mc, const MedBracket* __restrict__ br,
                                                    const int* __restrict__ count, const float* __restrict__ ref_all,
                                                    const float* __restrict__ tiles, const unsigned* __restrict__ vmask, int npix,
                                                    SelState* __restrict__ st, ColStat* __restrict__ cs) {
    __shared__ unsigned h[260];
    const int q = blockIdx.x, i = q / 20;
    const int n = count[i];
    const MedCount m = mc[q];
    const MedBracket b = br[q];
    const long long k = n > 0 ? (n - 1) / 2 : 0;
    const bool need_up = n > 0 && (n & 1) == 0;            // k_params_all reads the successor only for an even count
    const long long r = k - (long long)m.below;
    const int c = (int)min(m.ncand, (unsigned)kMedCap);
    const bool ok = n > 0 && !m.fail && b.lo <= b.hi && m.ncand <= (unsigned)kMedCap && r >= 0 && r + (need_up ? 1 : 0) < (long long)c;
    unsigned med = 0, up = 0;
    if (ok) {
        unsigned key[kMedCap / 256];                        // 96 candidates per thread, in registers
#pragma unroll
        for (int u = 0; u < kMedCap / 256; ++u) key[u] = (u * 256 + (int)threadIdx.x < c) ? cand[(long)q * kMedCap + u * 256 + threadIdx.x] : 0u;
        med = block_select_reg(key, c, r, h);
        up = need_up ? block_select_reg(key, c, r + 1, h) : med;
    } else if (n > 0) {                                    // the bracket missed (or overflowed): radix select over the full column
        const int which = q & 1, band = (q % 20) >> 1;
        const float* col = (which ? tiles : ref_all) + (long)i * npix * 10 + band;
        auto get = [&](int p, unsigned& kk) {
            if (!((vmask[p] >> i) & 1u)) return false;
            kk = fkey(col[(long)p * 10]);
            return true;
        };
        // ranks count VALID rows only: block_select skips the rows get() rejects
        med = block_select(get, npix, k, h);
        up = need_up ? block_select(get, npix, k + 1, h) : med;
        if (threadIdx.x == 0) mc[q].fell_back = 1u;
    }
    if (threadIdx.x == 0) {
        SelState ss; ss.prefix = med; ss.mask = 0xffffffffu; ss.k = 0;
        st[q] = ss;
        cs[q].up_key = up;
    }
}
__global__ void k_params_all(const SelState* __restrict__ st, const ColStat* __restrict__ cs, const int* __restrict__ count,
                             const int* __restrict__ n_land, int T, AlignPar* __restrict__ out) {
    const int i = threadIdx.x;
    if (i >= T) return;
    const int n = count[i];
    AlignPar ap;
    ap.ok = n > 1000;
    ap.any_land = *n_land > 0;
    float med[2], sd[2];
    for (int b = 0; b < 10; ++b) {
        for (int which = 0; which < 2; ++which) {
            const int q = i * 20 + b * 2 + which;
            const float lo = fkey_inv(st[q].prefix);
            // upper median = order statistic n / 2 = the one after the lower median of an even count (k_pick_last)
            float up = lo;
            if (n > 0 && (n & 1) == 0) up = fkey_inv(cs[q].up_key);
            med[which] = 0.5f * (lo + up);
            const double mean = cs[q].sum / (double)max(n, 1);
            double var = cs[q].sq / (double)max(n, 1) - mean * mean;
            if (var < 0.0) var = 0.0;
            sd[which] = (float)sqrt(var);
        }
        const float k = sd[0] / sd[1];                 // std_ref / std_src, CR.py:633
        ap.k[b] = k; ap.add[b] = med[0] - med[1] * k;
    }
    out[i] = ap;
}
// mosaic = sum_i (1 - w_i) * aligned_i (date order, CR.py:676), / divisor, NaN -> 10th percentile over dates, clamp
template <int TM>
__global__ void k_accum_final_all(const float* __restrict__ tiles, const float* __restrict__ w, const unsigned char* __restrict__ water,
                                  const AlignPar* __restrict__ ap, const float* __restrict__ divisor, int T, int npix,
                                  float* __restrict__ mosaic) {
#pragma clang fp contract(off)
    // thread = the float2 of two neighbouring bands of one pixel (10 bands: a pair never straddles pixels): 8-byte loads and stores.
    // The per-date alignment parameters come from LDS: read straight from `ap` they were four more vector loads per date and thread.
    __shared__ float2 sk[TM * 5], sadd[TM * 5];
    __shared__ int sok[TM];
    for (int e = threadIdx.x; e < T * 5; e += blockDim.x) {
        const int t = e / 5, c = 2 * (e - t * 5);
        sk[e] = float2{ap[t].k[c], ap[t].k[c + 1]}; sadd[e] = float2{ap[t].add[c], ap[t].add[c + 1]};
        if (c == 0) sok[t] = ap[t].ok ? 1 : 0;
    }
    __syncthreads();
    const long total2 = (long)npix * 5;
    const long id2 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id2 >= total2) return;
    const int p = (int)(id2 / 5), ch = 2 * (int)(id2 - (long)p * 5);
    const bool land = !water[p];
    float2 v[TM];
    float2 m = {0.f, 0.f}, mn = {INFINITY, INFINITY}, mx = {-INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        v[t] = float2{INFINITY, INFINITY};
        if (t < T) {
            v[t] = reinterpret_cast<const float2*>(tiles + (long)t * npix * 10)[id2];
            mn.x = fminf(mn.x, v[t].x); mx.x = fmaxf(mx.x, v[t].x);
            mn.y = fminf(mn.y, v[t].y); mx.y = fmaxf(mx.y, v[t].y);
            if (sok[t]) {
                const float wi = 1.0f - w[(long)t * npix + p];
                const float2 kk = sk[t * 5 + (ch >> 1)], ad = sadd[t * 5 + (ch >> 1)];
                const float a0 = land ? v[t].x * kk.x + ad.x : v[t].x;
                const float a1 = land ? v[t].y * kk.y + ad.y : v[t].y;
                m.x = m.x + wi * a0;
                m.y = m.y + wi * a1;
            }
        }
    }
    float d = divisor[p];
    if (d < 0.f) d = 0.f;
    float mm[2] = {m.x / d, m.y / d};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (isnan(mm[e])) {
            float vv[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t) vv[t] = e ? v[t].y : v[t].x;
            bitonic_sort<TM>(vv);
            const double pos = 0.1 * (T - 1);
            const int lo = (int)floor(pos);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int t = 0; t < TM; ++t) { if (t == lo) a = vv[t]; if (t == lo + 1 && lo + 1 < T) b = vv[t]; }
            if (lo + 1 >= T) b = a;
            mm[e] = (float)((double)a + ((double)b - (double)a) * (pos - lo));
        }
    }
    mm[0] = fminf(fmaxf(mm[0], mn.x), mx.x);
    mm[1] = fminf(fmaxf(mm[1], mn.y), mx.y);
    reinterpret_cast<float2*>(mosaic)[id2] = float2{mm[0], mm[1]};
}

__global__ void k_count_flags(const int* __restrict__ flags, int T, int* __restrict__ out) {
    if (threadIdx.x == 0) { int n = 0; for (int i = 0; i < T; ++i) n += flags[i] ? 1 : 0; out[0] = n; }
}

__global__ void k_redo_flag(const AlignPar* __restrict__ ap, int T, int* __restrict__ status) {
    if (threadIdx.x == 0) {
        int redo = 0;
        for (int i = 0; i < T; ++i) redo |= (!ap[i].ok && ap[i].any_land) ? 1 : 0;
        status[0] = redo;
    }
}

// water_raw (may be null): the undilated NDWI water mask of k_water<false>, already computed by the caller's fused pass (k_water_snow)
static ttc_status aligned_mosaic_impl(ttc_ctx* c, const float* d_tiles, float* d_w, int T, int X, int Y, float* d_mosaic,
                                      const unsigned char* water_raw, hipStream_t s);
ttc_status gapfill_aligned_mosaic(ttc_ctx* c, const float* d_tiles, float* d_w, int T, int X, int Y, float* d_mosaic,
                                  hipStream_t s) {
    return aligned_mosaic_impl(c, d_tiles, d_w, T, X, Y, d_mosaic, nullptr, s);
}
static ttc_status aligned_mosaic_impl(ttc_ctx* c, const float* d_tiles, float* d_w, int T, int X, int Y, float* d_mosaic,
                                      const unsigned char* water_raw, hipStream_t s) {
    if (!d_tiles || !d_w || !d_mosaic || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "aligned_mosaic: bad argument (T in [1,32])");
    const int npix = X * Y;
    const long total = (long)npix * 10;
    unsigned char* water = static_cast<unsigned char*>(c->scratch_buf("gf_water", (size_t)npix));
    float* ref_all = static_cast<float*>(c->scratch_buf("gf_ref_all", sizeof(float) * (size_t)T * total));
    unsigned* vmask = static_cast<unsigned*>(c->scratch_buf("gf_vmask", sizeof(unsigned) * (size_t)npix));
    float* divisor = static_cast<float*>(c->scratch_buf("gf_div", sizeof(float) * (size_t)npix));
    const size_t ctl_bytes = 8192 + sizeof(SelState) * kMaxT * 20 + sizeof(ColStat) * kMaxT * 20 + sizeof(AlignPar) * kMaxT +
                             sizeof(unsigned) * kMaxT * 20 * 256 + sizeof(int) * kMaxT + (sizeof(MedBracket) + sizeof(MedCount)) * kMaxT * 20;
    char* ctl = static_cast<char*>(c->scratch_buf("gf_ctl_all", ctl_bytes));
    if (!water || !ref_all || !vmask || !divisor || !ctl) return c->fail(TTC_ERR_NOMEM, "aligned_mosaic scratch");
    int* count = reinterpret_cast<int*>(ctl);                              // [kMaxT] valid rows per date, [kMaxT] = n_land
    SelState* st = reinterpret_cast<SelState*>(ctl + 8192);
    ColStat* cs = reinterpret_cast<ColStat*>(st + kMaxT * 20);
    AlignPar* ap = reinterpret_cast<AlignPar*>(cs + kMaxT * 20);
    unsigned* hist = reinterpret_cast<unsigned*>(ap + kMaxT);
    int* nsamp = reinterpret_cast<int*>(hist + kMaxT * 20 * 256);
    MedBracket* br = reinterpret_cast<MedBracket*>(nsamp + kMaxT);
    MedCount* mc = reinterpret_cast<MedCount*>(br + kMaxT * 20);
    // read per call (two getenv's per tile): the parity tests switch the three forms inside one process
    const bool med_radix = [] { const char* e = getenv("TTC_MEDIAN_RADIX"); return e && atoi(e) != 0; }();
    const int med_force = [] { const char* e = getenv("TTC_MEDIAN_FORCE_FALLBACK"); return e ? atoi(e) : 0; }();
    {
        KTimer kt(c, "aligned_mosaic", s);
        if (water_raw) {     // binary_dilation(1 - water, 2) then binary_dilation(1 - that, 5)  (CR.py:585-586), as in water_mask()
            unsigned char* tmp = static_cast<unsigned char*>(c->scratch_buf("gf_wtmp", 2 * (size_t)npix));
            if (!tmp) return c->fail(TTC_ERR_NOMEM, "water scratch");
            TTC_CHECK(dilate_diamond(c, water_raw, X, Y, 2, 1, tmp + npix, s));
            TTC_CHECK(dilate_diamond(c, tmp + npix, X, Y, 5, 1, water, s));
        } else TTC_CHECK(water_mask(c, d_tiles, T, X, Y, false, true, water, s));
        const dim3 grid((npix + 255) / 256), blk(256);
        TTC_HIP(c, hipMemsetAsync(ctl, 0, ctl_bytes, s));
        hipLaunchKernelGGL(k_divisor, grid, blk, 0, s, d_w, T, npix, divisor);
        hipLaunchKernelGGL(k_count_land, dim3(64), blk, 0, s, water, npix, count + kMaxT);
        GF_T(k_ref_all, T, dim3(2048), blk, 0, s, d_tiles, d_w, water, T, npix, ref_all, vmask, count);
        hipLaunchKernelGGL(k_sel_init_dates, dim3((T * 20 + 63) / 64), dim3(64), 0, s, st, count, T);
        hipLaunchKernelGGL(k_stat_init, dim3(1), dim3(kMaxT * 20), 0, s, cs);
        if (!med_radix) {
            // medians from a sampled bracket: one pass over the stack (see k_med_sample .. k_med_final)
            unsigned* samp = static_cast<unsigned*>(c->scratch_buf("gf_med_samp", sizeof(unsigned) * (size_t)T * 20 * kMedS));
            unsigned* cand = static_cast<unsigned*>(c->scratch_buf("gf_med_cand", sizeof(unsigned) * (size_t)T * 20 * kMedCap));
            if (!samp || !cand) return c->fail(TTC_ERR_NOMEM, "aligned_mosaic scratch");
            hipLaunchKernelGGL(k_med_sample, dim3(kMedS / 256, T), blk, 0, s, ref_all, d_tiles, vmask, npix, samp, nsamp);
            hipLaunchKernelGGL(k_med_bracket, dim3(T * 20), blk, 0, s, samp, nsamp, count, br, med_force);
            hipLaunchKernelGGL(k_med_count, dim3(128, T), blk, 0, s, ref_all, d_tiles, vmask, npix, br, mc, cand, cs);
            hipLaunchKernelGGL(k_med_final, dim3(T * 20), blk, 0, s, cand, mc, br, count, ref_all, d_tiles, vmask, npix, st, cs);
            hipLaunchKernelGGL(k_params_all, dim3(1), dim3(64), 0, s, st, cs, count, count + kMaxT, T, ap);
            TTC_HIP(c, hipGetLastError());
            c->named["gf_med_counts"] = {reinterpret_cast<float*>(mc), (size_t)T * 20 * 4};      // tests: candidates / fallbacks per problem
        } else {
        // 128 x T workgroups: measured 64 / 128 / 256 / 512 -> 3.58 / 3.47 / 3.51 / 3.66 ms per tile for the whole preprocessing chain
        hipLaunchKernelGGL(k_hist_all<1>, dim3(128, T), blk, 0, s, ref_all, d_tiles, vmask, npix, st, 24, hist, cs);      // + the moments
        hipLaunchKernelGGL(k_sel_pick, dim3(T * 20), dim3(64), 0, s, st, 24, hist);
        for (int shift = 16; shift >= 8; shift -= 8) {
            hipLaunchKernelGGL(k_hist_all<0>, dim3(128, T), blk, 0, s, ref_all, d_tiles, vmask, npix, st, shift, hist, cs);
            hipLaunchKernelGGL(k_sel_pick, dim3(T * 20), dim3(64), 0, s, st, shift, hist);
        }
        hipLaunchKernelGGL(k_hist_all<2>, dim3(128, T), blk, 0, s, ref_all, d_tiles, vmask, npix, st, 0, hist, cs);       // + the successor key
        hipLaunchKernelGGL(k_pick_last, dim3(T * 20), dim3(64), 0, s, st, hist, cs);
        hipLaunchKernelGGL(k_params_all, dim3(1), dim3(64), 0, s, st, cs, count, count + kMaxT, T, ap);
        TTC_HIP(c, hipGetLastError());
        }
    }
    // the one host decision: a date with <= 1000 usable rows on a tile that has land marks itself fully interpolated
    // and thereby changes every later date (CR.py:679-680) -> redo date by date.  T small ints, one stream wait.
    // Single-call tile path (c->spec_status set): no wait -- the batched result is used speculatively and the condition is
    // recorded in device memory, status[0] != 0 tells the caller afterwards that this tile needs the staged path.
    if (c->spec_status) {
        hipLaunchKernelGGL(k_redo_flag, dim3(1), dim3(64), 0, s, ap, T, c->spec_status);
        KTimer kt(c, "aligned_mosaic", s);
        GF_T(k_accum_final_all, T, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, s, d_tiles, d_w, water, ap, divisor, T, npix, d_mosaic);
        TTC_HIP(c, hipGetLastError());
        return TTC_OK;
    }
    AlignPar h_ap[kMaxT];
    TTC_HIP(c, hipMemcpyAsync(h_ap, ap, sizeof(AlignPar) * T, hipMemcpyDeviceToHost, s));
    TTC_HIP(c, hipStreamSynchronize(s));
    bool redo = false;
    for (int i = 0; i < T; ++i) redo |= (!h_ap[i].ok && h_ap[i].any_land);
    if (redo) return aligned_mosaic_sequential(c, d_tiles, d_w, T, X, Y, d_mosaic, s);
    KTimer kt(c, "aligned_mosaic", s);
    GF_T(k_accum_final_all, T, dim3((unsigned)((total / 2 + 255) / 256)), dim3(256), 0, s, d_tiles, d_w, water, ap, divisor, T, npix, d_mosaic);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// a8 + a9 -------------------------------------------------------------------------------------------
ttc_status gapfill_remove_clouds(ttc_ctx* c, float* d_tiles, const float* d_probs, const uint8_t* d_pfcps, int T, int X, int Y,
                                 ttc_sampler_fn sampler, void* user, float* d_interp, float* d_mosaic_out, int32_t* h_to_remove,
                                 int32_t* n_to_remove, hipStream_t s) {
    if (!d_tiles || !d_probs || !d_interp || T < 1 || T > kMaxT) return c->fail(TTC_ERR_ARG, "remove_cloud_and_shadows: bad argument (T in [1,32])");
    const int npix = X * Y;
    float* mosaic = d_mosaic_out ? d_mosaic_out : static_cast<float*>(c->scratch_buf("gf_mosaic", sizeof(float) * 10 * (size_t)npix));
    float* snow = static_cast<float*>(c->scratch_buf("gf_snow", sizeof(float) * (size_t)npix));
    unsigned char* water2 = static_cast<unsigned char*>(c->scratch_buf("gf_water2", (size_t)npix));
    unsigned char* bits = static_cast<unsigned char*>(c->scratch_buf("gf_bits", 4 * (size_t)npix));
    int* rows_all = static_cast<int*>(c->scratch_buf("gf_rows", sizeof(int) * 3 * (size_t)npix * T));
    float* snowp = static_cast<float*>(c->scratch_buf("gf_snowp", sizeof(float) * (size_t)npix * T));
    float* evi = static_cast<float*>(c->scratch_buf("gf_evi", sizeof(float) * 3 * (size_t)npix));
    const int nblk_rows = (3 * npix + 255) / 256;
    int* blk = static_cast<int*>(c->scratch_buf("gf_blk", sizeof(int) * ((size_t)nblk_rows * T + 16 + 6 * kMaxT)));
    const int gram_blocks = 1024;
    double* gpart = static_cast<double*>(c->scratch_buf("gf_gram", sizeof(double) * 1024 * (gram_blocks + 1)));
    if (!mosaic || !snow || !water2 || !bits || !rows_all || !snowp || !evi || !blk || !gpart)
        return c->fail(TTC_ERR_NOMEM, "gap-fill scratch");
    int* counters = blk + (size_t)nblk_rows * T;                         // a9: [3] total rows, [4] n_only
    int* date_counts = counters + 16;                                    // [T][4]: n(w > 0), n(w == 0), n(w < 1)
    int* screen = date_counts + 4 * kMaxT;                               // [T][2]: snow-flag pixels, missing pixels (single-call path)
    if (n_to_remove) *n_to_remove = 0;
    const dim3 grid((npix + 255) / 256), b256(256);

    // the stack is read-only until the date loop: the aligned mosaic's NDWI water mask (CR.py:580-584), the median water mask of CR.py:936-939,
    // the per-date snow probabilities (CR.py:348-372) and the screening counts of the single-call path all come out of ONE pass over it
    unsigned char* water_raw = static_cast<unsigned char*>(c->scratch_buf("gf_water_raw", (size_t)npix));
    if (!water_raw) return c->fail(TTC_ERR_NOMEM, "gap-fill scratch");
    TTC_HIP(c, hipMemsetAsync(date_counts, 0, sizeof(int) * 6 * kMaxT, s));
    {
        KTimer kt0(c, "water_snow", s);
        GF_T(k_water_snow, T, grid, b256, 0, s, d_tiles, T, npix, water_raw, water2, snowp, c->spec_status ? screen : nullptr);
    }
    TTC_CHECK(gapfill_feather(c, d_probs, T, X, Y, 20, 0, d_interp, s));                       // CR.py:910-923
    TTC_CHECK(aligned_mosaic_impl(c, d_tiles, d_interp, T, X, Y, mosaic, water_raw, s));       // CR.py:925
    char* ctl = static_cast<char*>(c->scratch_buf("gf_ctl", 65536));
    if (!ctl) return c->fail(TTC_ERR_NOMEM, "gap-fill control block");
    DatePlan* plans = reinterpret_cast<DatePlan*>(ctl + 32768);            // [kMaxT]
    Beta* d_beta = reinterpret_cast<Beta*>(ctl + 64);                      // 888 B
    SelState* st = reinterpret_cast<SelState*>(ctl + 2048);                // 12 problems
    float* thr = reinterpret_cast<float*>(ctl + 3072);
    int* remove_flags = reinterpret_cast<int*>(ctl + 3200);                // kMaxT ints
    unsigned* hist = reinterpret_cast<unsigned*>(ctl + 4096);              // 12 * 256 * 4 = 12288 B
    double* Zdev = gpart + 1024L * gram_blocks;
    TTC_HIP(c, hipMemsetAsync(ctl, 0, 4096 + 12288, s));
    TTC_HIP(c, hipMemsetAsync(date_counts, 0, sizeof(int) * 4 * kMaxT, s));      // [T][4]; the screening counts behind them were filled by k_water_snow
    const PctList pl6{{2, 20, 40, 60, 80, 98, 0, 0}};
    const int nb3 = (int)((3L * npix + 255) / 256);

    KTimer kt(c, "gapfill_dates", s);
    // date-loop invariants, all dates at once: clear-pixel counts -> plans (which dates train which fit) -> row lists
    int* const spec = c->spec_status;            // single-call tile path: report instead of deciding on the host
    hipLaunchKernelGGL(k_date_counts_all, dim3(32, T), b256, 0, s, d_interp, npix, date_counts);
    hipLaunchKernelGGL(k_date_plan, dim3(1), dim3(64), 0, s, date_counts, npix, X, T, plans, remove_flags, screen, spec);
    hipLaunchKernelGGL(k_rows_count, dim3(nb3, T), b256, 0, s, d_interp, water2, npix, 0, 0, plans, blk);
    hipLaunchKernelGGL(k_rows_scan, dim3(T), dim3(1024), 0, s, blk, nb3, &plans->nrows, (int)(sizeof(DatePlan) / sizeof(int)));
    if (sampler) hipLaunchKernelGGL(k_rows_fill, dim3(nb3, T), b256, 0, s, d_interp, water2, d_tiles, npix, 0, 0, plans, blk, rows_all, (float*)nullptr);
    TTC_HIP(c, hipGetLastError());
    if (!sampler) {
        // device sampler: everything that does not depend on the blended stack for all dates at once (see k_gram_snow)
        constexpr int kGramBlocks = 256;
        float* evi_all = static_cast<float*>(c->scratch_buf("gf_evi_all", sizeof(float) * 3 * (size_t)npix * T));
        float* weight_all = static_cast<float*>(c->scratch_buf("gf_weight_all", sizeof(float) * 3 * (size_t)npix * T));
        double* gpart_all = static_cast<double*>(c->scratch_buf("gf_gram_all", sizeof(double) * 1024 * ((size_t)kGramBlocks * T + T) +
                                                                                sizeof(double) * 32 * (kSnowBlocks + 1)));
        char* ctl2 = static_cast<char*>(c->scratch_buf("gf_ctl_dates", sizeof(SelState) * 12 * kMaxT + sizeof(StrataDev) * kMaxT +
                                                                       sizeof(unsigned) * (12 * 256 * kMaxT + 16)));
        if (!evi_all || !weight_all || !gpart_all || !ctl2) return c->fail(TTC_ERR_NOMEM, "gap-fill scratch (date batch)");
        double* Z0 = gpart_all + 1024L * kGramBlocks * T;              // [T][32][32]
        double* spart = Z0 + 1024L * T;                                // [kSnowBlocks][32], then the reduced [32]
        SelState* st_all = reinterpret_cast<SelState*>(ctl2);
        StrataDev* sd_all = reinterpret_cast<StrataDev*>(st_all + 12 * kMaxT);
        unsigned* hist_all = reinterpret_cast<unsigned*>(sd_all + kMaxT);
        TTC_HIP(c, hipMemsetAsync(hist_all, 0, sizeof(unsigned) * (12 * 256 * kMaxT + 16), s));
        hipLaunchKernelGGL(k_rows_fill, dim3(nb3, T), b256, 0, s, d_interp, water2, d_tiles, npix, 0, 0, plans, blk, rows_all, evi_all);
        hipLaunchKernelGGL(k_sel_init_evi, dim3((T * 12 + 63) / 64), dim3(64), 0, s, st_all, plans, T, pl6);
        for (int shift = 24; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(k_evi_hist, dim3(64, T), b256, 0, s, evi_all, plans, npix, st_all, shift, hist_all);
            hipLaunchKernelGGL(k_sel_pick, dim3(T * 12), dim3(64), 0, s, st_all, shift, hist_all);
        }
        hipLaunchKernelGGL(k_strata_thresholds_all, dim3(1), dim3(64), 0, s, st_all, plans, T, pl6, sd_all);
        hipLaunchKernelGGL(k_strata_count_all, dim3(64, T), b256, 0, s, evi_all, plans, npix, sd_all);
        hipLaunchKernelGGL(k_row_weights_all, dim3(128, T), b256, 0, s, evi_all, plans, npix, sd_all, weight_all);
        hipLaunchKernelGGL(k_snow_mean_cached, grid, b256, 0, s, snowp, T, npix, snow);       // CR.py:372; k_predict_blend keeps it current
        hipLaunchKernelGGL(k_gram_all, dim3(kGramBlocks, T), b256, 0, s, d_tiles, mosaic, rows_all, weight_all, plans, npix, gpart_all);
        hipLaunchKernelGGL(k_gram_reduce_all, dim3(64, T), b256, 0, s, gpart_all, kGramBlocks, Z0);
        TTC_HIP(c, hipGetLastError());
        for (int date = 0; date < T; ++date) {
            DatePlan* plan = plans + date;
            hipLaunchKernelGGL(k_gram_snow, dim3(kSnowBlocks), dim3(1024), 0, s, d_tiles, mosaic, snow, rows_all + (size_t)date * 3 * npix,
                               weight_all + (size_t)date * 3 * npix, plan, npix, spart);
            hipLaunchKernelGGL(k_nnls, dim3(10), dim3(64), 0, s, Z0 + 1024L * date, plan, d_beta, spart, kSnowBlocks);
            hipLaunchKernelGGL(k_predict_blend, grid, b256, 0, s, d_tiles, d_interp, mosaic, snow, d_beta, npix, date, snowp, T,
                               spec ? 1 : 0);
        }
        TTC_HIP(c, hipGetLastError());
    }
    std::vector<float> h_evi;
    std::vector<int64_t> h_idx;
    for (int date = 0; sampler && date < T; ++date) {
        DatePlan* plan = plans + date;
        const int* rows = rows_all + (size_t)date * 3 * npix;
        hipLaunchKernelGGL(k_snow_mean_cached, grid, b256, 0, s, snowp, T, npix, snow);       // CR.py:372 (tiles mutate per date)
        hipLaunchKernelGGL(k_rows_evi, dim3(256), b256, 0, s, rows, plan, d_tiles, npix, evi);
        GramArgs ga{d_tiles, mosaic, snow, rows, nullptr, nullptr, 0, npix, 0, plan};
        {   // reference replay: host round trip through the callback, which returns row indices
            DatePlan hp;
            TTC_HIP(c, hipMemcpyAsync(&hp, plan, sizeof(hp), hipMemcpyDeviceToHost, s));
            TTC_HIP(c, hipStreamSynchronize(s));
            if (hp.proceed && hp.nrows > 0) {
                h_evi.resize(hp.nrows);
                TTC_HIP(c, hipMemcpyAsync(h_evi.data(), evi, sizeof(float) * hp.nrows, hipMemcpyDeviceToHost, s));
                TTC_HIP(c, hipStreamSynchronize(s));
                h_idx.resize((size_t)hp.nrows * 3 + 16);
                const int64_t ns = sampler(h_evi.data(), hp.nrows, h_idx.data(), (int64_t)h_idx.size(), user);
                if (ns <= 0 || ns > (int64_t)h_idx.size()) return c->fail(TTC_ERR_ARG, "sampler callback returned a bad count");
                std::vector<int> idx32(ns);
                for (int64_t i = 0; i < ns; ++i) {
                    if (h_idx[i] < 0 || h_idx[i] >= hp.nrows) return c->fail(TTC_ERR_ARG, "sampler callback returned an out-of-range row");
                    idx32[i] = (int)h_idx[i];
                }
                int* d_sample = static_cast<int*>(c->scratch_buf("gf_sample", sizeof(int) * (size_t)ns));
                if (!d_sample) return c->fail(TTC_ERR_NOMEM, "sample buffer");
                TTC_HIP(c, hipMemcpyAsync(d_sample, idx32.data(), sizeof(int) * ns, hipMemcpyHostToDevice, s));
                TTC_HIP(c, hipStreamSynchronize(s));
                ga.sample = d_sample; ga.nsample = ns; ga.t0 = hp.t0; ga.plan = nullptr;
            }
        }
        hipLaunchKernelGGL(k_gram, dim3(gram_blocks), b256, 0, s, ga, gpart);
        hipLaunchKernelGGL(k_gram_reduce, dim3(64), b256, 0, s, gpart, gram_blocks, Zdev);
        hipLaunchKernelGGL(k_nnls, dim3(10), dim3(64), 0, s, Zdev, plan, d_beta);               // 10 fits of 11 unknowns
        hipLaunchKernelGGL(k_predict_blend, grid, b256, 0, s, d_tiles, d_interp, mosaic, snow, d_beta, npix, date, snowp);
        TTC_HIP(c, hipGetLastError());
    }
    // a9: clouds that survive in the mosaic (CR.py:964-968)
    unsigned char *only1 = bits, *pfd = bits + npix, *cl = bits + 2 * (size_t)npix, *tmp = bits + 3 * (size_t)npix;
    if (d_pfcps) TTC_CHECK(dilate_diamond(c, d_pfcps, X, Y, 10, 0, pfd, s));
    else TTC_HIP(c, hipMemsetAsync(pfd, 0, npix, s));
    TTC_HIP(c, hipMemsetAsync(counters, 0, sizeof(int) * 4, s));
    hipLaunchKernelGGL(k_only1, grid, b256, 0, s, d_interp, pfd, T, npix, only1, counters);
    const PctList pl99{{99, 99, 0, 0, 0, 0, 0, 0}};
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(64), 0, s, st, 4, counters, npix, 1, 1, pl99);
    TTC_HIP(c, radix_select(SrcBlueRed{mosaic, only1, npix}, st, hist, 4, s));
    hipLaunchKernelGGL(k_cloud_thresholds, dim3(1), dim3(64), 0, s, st, counters, npix, thr);
    hipLaunchKernelGGL(k_cloud_flags, grid, b256, 0, s, mosaic, only1, pfd, thr, npix, cl);
    TTC_CHECK(dilate_diamond(c, cl, X, Y, 3, 1, tmp, s));              // dilate(1 - c, 3)
    TTC_CHECK(dilate_diamond(c, tmp, X, Y, 8, 1, cl, s));               // dilate(1 - that, 8)
    hipLaunchKernelGGL(k_add_clouds, grid, b256, 0, s, d_interp, cl, T, npix);
    TTC_HIP(c, hipGetLastError());
    if (c->spec_status) hipLaunchKernelGGL(k_count_flags, dim3(1), dim3(64), 0, s, remove_flags, T, c->spec_status + 2);
    if (h_to_remove && n_to_remove) {          // the only host read-back; skipped when the caller does not ask
        int flags[kMaxT];
        TTC_HIP(c, hipMemcpyAsync(flags, remove_flags, sizeof(int) * T, hipMemcpyDeviceToHost, s));
        TTC_HIP(c, hipStreamSynchronize(s));
        for (int t = 0; t < T; ++t) if (flags[t]) h_to_remove[(*n_to_remove)++] = t;
    }
    return TTC_OK;
}
