// Gaussian-weighted overlap mosaic of the per-window predictions:
// load_mosaic_predictions(out_folder, depth=1), src/download_and_predict_job.py:1515-1641,
// with fspecial_gauss (:1489-1501) and calc_overlap (:1503-1512), in gather form -- every output
// pixel visits the (<= 4 of 36) windows that cover it instead of materialising the reference's
// dense [618, 618, 36] NaN stacks.
//
// Orientation: the reference saves window (folder_x, folder_y) as processed/{folder_y}/{folder_x}.npy,
// loads it TRANSPOSED (job.py:1578) and places it at rows folder_y.., cols folder_x.. -- the result
// is [Y][X] relative to the [X][Y] tile arrays.  out[r][c] = window[c - folder_x][r - folder_y].
#include "ttc_internal.h"

namespace {

constexpr int kMaxWin = 64;
struct MWin { int n; int fx[kMaxWin], fy[kMaxWin]; };

// value as load_mosaic_predictions sees it after `prediction[prediction < 255] *= 100` (job.py:1576)
__device__ __forceinline__ float pct(float v) { return v < 255.0f ? v * 100.0f : v; }

// 1) per window: sum of the scaled values (placement test, job.py:1577)
__global__ void k_mos_sum(const float* __restrict__ win, int size, double* __restrict__ sums) {
    __shared__ double part[16];
    const int w = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < size * size; i += blockDim.x) s += (double)pct(win[(long)w * size * size + i]);
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
        sums[w] = t;
    }
}

// 2) calc_overlap: mean |nanmean(other windows) - this window| over this window's footprint
__global__ void k_mos_ratio(const float* __restrict__ win, MWin mw, int size, const double* __restrict__ sums,
                            float* __restrict__ ratios) {
    __shared__ double ps[16];
    __shared__ int pc[16];
    __shared__ int nbr[kMaxWin], n_nbr;
    const int w = blockIdx.x;
    const double thr = (double)size * size * 255.0;
    if (threadIdx.x == 0) {                                   // windows whose footprint intersects this one, in index order
        int k = 0;
        for (int j = 0; j < mw.n; ++j)
            if (j != w && sums[j] < thr && abs(mw.fx[j] - mw.fx[w]) < size && abs(mw.fy[j] - mw.fy[w]) < size) nbr[k++] = j;
        n_nbr = k;
    }
    __syncthreads();
    double acc = 0.0; int cnt = 0;
    if (sums[w] < thr) {
        for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
            const int r = i / size, c = i % size;            // transposed local coords (row ~ y, col ~ x)
            const float mine = pct(win[(long)w * size * size + (long)c * size + r]);
            const int R = mw.fy[w] + r, C = mw.fx[w] + c;
            float os = 0.f; int on = 0;
            for (int q = 0; q < n_nbr; ++q) {
                const int j = nbr[q];
                const int rr = R - mw.fy[j], cc = C - mw.fx[j];
                if (rr >= 0 && rr < size && cc >= 0 && cc < size) { os += pct(win[(long)j * size * size + (long)cc * size + rr]); ++on; }
            }
            if (on) { acc += (double)fabsf(os / (float)on - mine); ++cnt; }
        }
    }
    for (int k = 32; k >= 1; k >>= 1) { acc += __shfl_xor(acc, k); cnt += __shfl_xor(cnt, k); }
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = acc; pc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0; int n = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { t += ps[i]; n += pc[i]; }
        ratios[w] = n ? (float)(t / n) : NAN;
    }
}

// 3) multipliers = min(median(ratios) / ratios, 1.5) (job.py:1603-1606); all 1 when any window is
//    unplaced -- the reference's calc_overlap raises on it and the try/except skips the weighting.
__global__ void k_mos_mult(const float* __restrict__ ratios, const double* __restrict__ sums, int n, int size,
                           float* __restrict__ mult) {
    __shared__ float r[kMaxWin];
    __shared__ int bad;
    const int t = threadIdx.x;
    if (t == 0) bad = 0;
    __syncthreads();
    if (t < n) { r[t] = ratios[t]; if (!(sums[t] < (double)size * size * 255.0)) atomicOr(&bad, 1); }
    __syncthreads();
    if (t == 0) {
        float med = 0.f;
        if (!bad) {
            float s[kMaxWin];
            for (int i = 0; i < n; ++i) s[i] = r[i];
            for (int i = 1; i < n; ++i) { float v = s[i]; int j = i - 1; while (j >= 0 && s[j] > v) { s[j + 1] = s[j]; --j; } s[j + 1] = v; }
            med = (s[(n - 1) / 2] + s[n / 2]) * 0.5f;
        }
        for (int i = 0; i < n; ++i) {
            float m = 1.0f;
            if (!bad) { m = med / r[i]; if (m > 1.5f) m = 1.5f; }
            mult[i] = m;
        }
    }
}

// 4) blend (job.py:1580-1623)
__global__ void k_mos_blend(const float* __restrict__ win, MWin mw, int size, int rows, int cols,
                            const double* __restrict__ sums, const float* __restrict__ mult, double inv2s2,
                            unsigned char* __restrict__ u8, float* __restrict__ f32) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int R = i / cols, C = i % cols;
    const double thr = (double)size * size * 255.0;
    const int half = size / 2 - 1;                            // mgrid[-size//2+1 : size//2+1] -> index - (size//2 - 1) for even size
    float wsum = 0.f;
    int np = 0;
    for (int pass = 0; pass < 2; ++pass) {
        float acc = 0.f;
        for (int j = 0; j < mw.n; ++j) {
            if (!(sums[j] < thr)) continue;
            const int rr = R - mw.fy[j], cc = C - mw.fx[j];
            if (rr < 0 || rr >= size || cc < 0 || cc >= size) continue;
            const float p = pct(win[(long)j * size * size + (long)cc * size + rr]);
            if (p > 100.0f) continue;                          // no-data: weight 0, value NaN
            const int a = rr - half, b = cc - half;
            const float g = (float)exp(-((double)(a * a + b * b)) * inv2s2) * mult[j];
            if (pass == 0) { wsum += g; ++np; } else acc += p * (g / wsum);
        }
        if (pass == 1) wsum = acc;                             // reuse as the blended value
    }
    float out = np > 0 ? wsum : NAN;
    if (f32) f32[i] = out;
    unsigned char q;
    if (isnan(out)) q = 255;
    else {
        q = (unsigned char)out;                               // astype(uint8) truncation
        if (q <= 15) q = 0;
        if (q > 100) q = 255;
    }
    u8[i] = q;
}

// 5) no-data dilation, 10 iterations 8-connected == 21x21 max, separable (job.py:1636-1640)
__global__ void k_mos_dil_rows(const unsigned char* __restrict__ u8, int rows, int cols, unsigned char* __restrict__ tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int R = i / cols, C = i % cols;
    bool v = false;
    for (int d = -10; d <= 10; ++d) { const int c = C + d; if (c >= 0 && c < cols) v |= u8[R * cols + c] == 255; }
    tmp[i] = v;
}
__global__ void k_mos_dil_cols(const unsigned char* __restrict__ tmp, int rows, int cols, unsigned char* __restrict__ u8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int R = i / cols, C = i % cols;
    bool v = false;
    for (int d = -10; d <= 10; ++d) { const int r = R + d; if (r >= 0 && r < rows) v |= tmp[r * cols + C] != 0; }
    if (v) u8[i] = 255;
}

// feature mosaic, depth > 1 (job.py:1552-1592): plain Gaussian blend of the int16 feature windows, no no-data logic;
// uncovered pixels give 0 (nansum of nothing).  feats [n][size][size][depth], out [depth][rows][cols].
__global__ void k_mos_feats(const short* __restrict__ feats, MWin mw, int size, int depth, int rows, int cols, double inv2s2,
                            short* __restrict__ out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int R = i / cols, C = i % cols;
    const int half = size / 2 - 1;
    int wj[8];
    float wg[8];
    int nw = 0;
    float wsum = 0.f;
    for (int j = 0; j < mw.n && nw < 8; ++j) {
        const int rr = R - mw.fy[j], cc = C - mw.fx[j];
        if (rr < 0 || rr >= size || cc < 0 || cc >= size) continue;
        const int a = rr - half, b = cc - half;
        wg[nw] = (float)exp(-((double)(a * a + b * b)) * inv2s2);
        wj[nw] = (j * size + cc) * size + rr;                  // window j, element [cc][rr]
        wsum += wg[nw];
        ++nw;
    }
    for (int k = 0; k < nw; ++k) wg[k] = wg[k] / wsum;
    for (int d = 0; d < depth; ++d) {
        float acc = 0.f;
        for (int k = 0; k < nw; ++k) acc += (float)feats[(long)wj[k] * depth + d] * wg[k];
        out[(long)d * rows * cols + i] = (short)acc;
    }
}

}  // namespace

ttc_status mosaic_features(ttc_ctx* c, const int16_t* d_feats, int n, const int32_t* h_xy, int size, int depth, int rows, int cols,
                           int16_t* d_out, hipStream_t s) {
    if (!d_feats || !h_xy || !d_out) return c->fail(TTC_ERR_ARG, "mosaic_features: null argument");
    if (n < 1 || n > kMaxWin || depth < 1) return c->fail(TTC_ERR_ARG, "mosaic_features: window count must be in [1, 64], depth >= 1");
    if (size % 2 != 0) return c->fail(TTC_ERR_ARG, "mosaic_features: window size must be even");
    MWin mw{};
    mw.n = n;
    for (int i = 0; i < n; ++i) {
        mw.fx[i] = h_xy[2 * i]; mw.fy[i] = h_xy[2 * i + 1];
        if (mw.fx[i] < 0 || mw.fy[i] < 0 || mw.fx[i] + size > cols || mw.fy[i] + size > rows)
            return c->fail(TTC_ERR_ARG, "mosaic_features: window outside the output raster");
    }
    KTimer kt(c, "mosaic_features", s);
    const int np = rows * cols;
    hipLaunchKernelGGL(k_mos_feats, dim3((np + 255) / 256), dim3(256), 0, s, d_feats, mw, size, depth, rows, cols,
                       1.0 / (2.0 * 36.0 * 36.0), d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status mosaic_run(ttc_ctx* c, const float* d_windows, int n, const int32_t* h_xy, int size, int rows, int cols,
                      uint8_t* d_u8, float* d_f32, hipStream_t s) {
    if (!d_windows || !h_xy || !d_u8) return c->fail(TTC_ERR_ARG, "mosaic: null argument");
    if (n < 1 || n > kMaxWin) return c->fail(TTC_ERR_ARG, "mosaic: window count must be in [1, 64]");
    if (size % 2 != 0) return c->fail(TTC_ERR_ARG, "mosaic: window size must be even");
    MWin mw{};
    mw.n = n;
    for (int i = 0; i < n; ++i) {
        mw.fx[i] = h_xy[2 * i]; mw.fy[i] = h_xy[2 * i + 1];
        if (mw.fx[i] < 0 || mw.fy[i] < 0 || mw.fx[i] + size > cols || mw.fy[i] + size > rows)
            return c->fail(TTC_ERR_ARG, "mosaic: window outside the output raster");
    }
    double* sums = static_cast<double*>(c->scratch_buf("mos_sums", sizeof(double) * kMaxWin));
    float* ratios = static_cast<float*>(c->scratch_buf("mos_ratio", sizeof(float) * 2 * kMaxWin));
    unsigned char* tmp = static_cast<unsigned char*>(c->scratch_buf("mos_tmp", (size_t)rows * cols));
    if (!sums || !ratios || !tmp) return c->fail(TTC_ERR_NOMEM, "mosaic scratch");
    float* mult = ratios + kMaxWin;
    c->named["mos_ratios"] = {ratios, (size_t)n};
    c->named["mos_mult"] = {mult, (size_t)n};
    KTimer kt(c, "mosaic", s);
    hipLaunchKernelGGL(k_mos_sum, dim3(n), dim3(256), 0, s, d_windows, size, sums);
    hipLaunchKernelGGL(k_mos_ratio, dim3(n), dim3(1024), 0, s, d_windows, mw, size, sums, ratios);
    hipLaunchKernelGGL(k_mos_mult, dim3(1), dim3(64), 0, s, ratios, sums, n, size, mult);
    const int np = rows * cols;
    const double sigma = 36.0;
    hipLaunchKernelGGL(k_mos_blend, dim3((np + 255) / 256), dim3(256), 0, s, d_windows, mw, size, rows, cols, sums, mult,
                       1.0 / (2.0 * sigma * sigma), d_u8, d_f32);
    hipLaunchKernelGGL(k_mos_dil_rows, dim3((np + 255) / 256), dim3(256), 0, s, d_u8, rows, cols, tmp);
    hipLaunchKernelGGL(k_mos_dil_cols, dim3((np + 255) / 256), dim3(256), 0, s, tmp, rows, cols, d_u8);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
