// Tile-border re-prediction: the numeric core of `process_subtiles` in src/resegment_tiles_wide.py:360-616.
//
// A border strip is the last SIZE/2+7 columns of a tile next to the first SIZE/2+7 columns of its right-hand
// neighbour ([12, X, SIZE+14, 14] smoothed bands + indices, :84-115); four [SIZE_Y+14, SIZE+14] windows stacked along
// the strip are re-predicted with the non-square graph (ttc_config.win_rows).  Everything below is HBM-streaming:
//   k_steps_medians    NaN fix (interpolation.py:42-56), median over the 12 steps, quarterly medians (:397-408)
//   k_hist_stats       per window / frame / half: masked band sums for align_subtile_histograms (:284-343)
//   k_hist_decide      the affine per band and half, kept only when the seam step shrinks (:323-341)
//   k_border_assemble  window cut + 7-row reflect pad (:451-474) + 17-channel assembly (:476-490) + float32
//                      normalisation (:199-200), written straight into the model's padded planar frames
//   k_seam_adjust      pulls the two halves of a prediction together (:518-531) + the scalars the host's keep / skip
//                      decision needs (:534-613)
#include "ttc_internal.h"

namespace {

constexpr int kMaxBW = 8;
struct BWin { int start, h, pad0; };
struct BTable { int n; BWin w[kMaxBW]; };
struct Norm17 { float lo[17], hi[17], mid[17], half[17]; };

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ void cswap(float& a, float& b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }
__device__ __forceinline__ float med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }

// in [12][n] -> q [4][n] (median of each 3 consecutive steps), med [n] (median of 12 = mean of the two middle values)
__global__ void k_steps_medians(const float* __restrict__ in, long n, int nanfix, float* __restrict__ q, float* __restrict__ med) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        v[t] = in[(long)t * n + e];
        if (nanfix && v[t] != v[t]) v[t] = 0.0f;       // NaN -> (NaN-propagating median -> 0), interpolation.py:47-51
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) q[(long)k * n + e] = med3(v[3 * k], v[3 * k + 1], v[3 * k + 2]);
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 11 - i; ++j) cswap(v[j], v[j + 1]);
    med[e] = (v[5] + v[6]) * 0.5f;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// stats [(w*5 + f)*2 + side][14][3] = (count, sum, sum of squares) over the non-water, non-NaN pixels; side 0 = columns
// [:half], side 1 = columns [half:].  Water: NDWI of the median over the window's frames >= 0.1 (:286-298).
__global__ __launch_bounds__(256) void k_hist_stats(const float* __restrict__ q, const float* __restrict__ med, BTable bt, int W,
                                                     long npix, int half, double* __restrict__ stats) {
#pragma clang fp contract(off)
    const int wf = blockIdx.x, w = wf / 5, f = wf - 5 * w, side = blockIdx.y;
    const BWin bw = bt.w[w];
    const int c0 = side ? half : 0, nc = side ? W - half : half;
    const int rows_per = (bw.h + gridDim.z - 1) / gridDim.z;
    const int r0 = blockIdx.z * rows_per, r1 = min(bw.h, r0 + rows_per);
    double acc[14][3];
    for (int b = 0; b < 14; ++b) acc[b][0] = acc[b][1] = acc[b][2] = 0.0;
    for (int i = r0 * nc + threadIdx.x; i < r1 * nc; i += blockDim.x) {
        const int r = i / nc, cc = i - r * nc;
        const long tp = (long)(bw.start + r) * W + c0 + cc;
        float g, n;
        if (f < 4) {
            float a[4], b[4];
            for (int k = 0; k < 4; ++k) { a[k] = q[((long)k * npix + tp) * 14 + 1]; b[k] = q[((long)k * npix + tp) * 14 + 3]; }
            cswap(a[0], a[1]); cswap(a[2], a[3]); cswap(a[0], a[2]); cswap(a[1], a[3]); cswap(a[1], a[2]);
            cswap(b[0], b[1]); cswap(b[2], b[3]); cswap(b[0], b[2]); cswap(b[1], b[3]); cswap(b[1], b[2]);
            g = (a[1] + a[2]) * 0.5f; n = (b[1] + b[2]) * 0.5f;
        } else { g = med[tp * 14 + 1]; n = med[tp * 14 + 3]; }
        const bool water = ((g - n) / (g + n)) >= 0.1f;
        if (water) continue;
        const float* src = f < 4 ? q + ((long)f * npix + tp) * 14 : med + tp * 14;
        for (int b = 0; b < 14; ++b) {
            const float v = src[b];
            if (v == v) { acc[b][0] += 1.0; acc[b][1] += (double)v; acc[b][2] += (double)v * (double)v; }
        }
    }
    // one partial per row chunk, summed in chunk order by k_hist_decide: the result does not depend on scheduling
    __shared__ double red[4];
    double* out = stats + (((long)wf * 2 + side) * gridDim.z + blockIdx.z) * 42;
    for (int b = 0; b < 14; ++b)
        for (int k = 0; k < 3; ++k) {
            const double v = block_sum(acc[b][k], red);
            if (threadIdx.x == 0) out[b * 3 + k] = v;
        }
}

// aff [(w*5 + f)][side][14][2] = (mult, add) applied to columns of that side; (1, 0) when the alignment is rejected
__global__ __launch_bounds__(256) void k_hist_decide(const float* __restrict__ q, const float* __restrict__ med, BTable bt, int W,
                                                      long npix, int half, int seam_col, const double* __restrict__ stats, int nchunk,
                                                      float* __restrict__ aff, int* __restrict__ applied) {
#pragma clang fp contract(off)
    __shared__ double red[4];
    __shared__ float ma[2][14][2];
    const int wf = blockIdx.x, w = wf / 5, f = wf - 5 * w;
    const BWin bw = bt.w[w];
    if (threadIdx.x < 14) {
        const int b = threadIdx.x;
        float mean[2], sd[2];
        for (int side = 0; side < 2; ++side) {
            double s[3] = {0.0, 0.0, 0.0};
            for (int ch = 0; ch < nchunk; ++ch) {
                const double* p = stats + (((long)wf * 2 + side) * nchunk + ch) * 42 + b * 3;
                s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
            }
            const double m = s[1] / s[0];
            double var = s[2] / s[0] - m * m;
            if (var < 0) var = 0;
            mean[side] = (float)m; sd[side] = (float)sqrt(var);
        }
        // reference names: `right` = columns [:half] (side 0), `left` = columns [half:] (side 1); columns [:half] are
        // rescaled with the LEFT statistics and vice versa (:315-326)
        const float std_ref = (sd[0] + sd[1]) / 2.0f, mean_ref = (mean[0] + mean[1]) / 2.0f;
        const float mult_l = sd[1] / std_ref, add_l = mean[1] - mean_ref * mult_l;
        const float mult_r = sd[0] / std_ref, add_r = mean[0] - mean_ref * mult_r;
        ma[0][b][0] = mult_l; ma[0][b][1] = add_l;
        ma[1][b][0] = mult_r; ma[1][b][1] = add_r;
    }
    __syncthreads();
    double before = 0.0, after = 0.0;
    for (int i = threadIdx.x; i < bw.h * 14; i += blockDim.x) {
        const int r = i / 14, b = i - r * 14;
        const long tp = (long)(bw.start + r) * W + seam_col;
        const float* src = f < 4 ? q + ((long)f * npix + tp) * 14 : med + tp * 14;
        const float x0 = src[b - 14], x1 = src[b];           // columns seam_col - 1 and seam_col
        before += (double)fabsf(x0 - x1);
        const float c0 = x0 * ma[seam_col - 1 >= half][b][0] + ma[seam_col - 1 >= half][b][1];
        const float c1 = x1 * ma[seam_col >= half][b][0] + ma[seam_col >= half][b][1];
        after += (double)fabsf(c0 - c1);
    }
    before = block_sum(before, red);
    after = block_sum(after, red);
    const bool take = after < before;                         // NaN statistics -> rejected, as `after < before` is False
    if (threadIdx.x < 28) {
        const int side = threadIdx.x / 14, b = threadIdx.x - 14 * side;
        float* o = aff + (((long)wf * 2 + side) * 14 + b) * 2;
        o[0] = take ? ma[side][b][0] : 1.0f;
        o[1] = take ? ma[side][b][1] : 0.0f;
    }
    if (threadIdx.x == 0) applied[wf] = take ? 1 : 0;
}

// frames [n][5][17][plane]; frames 0..3 reflect-padded, frame 4 (medians) zero-padded, as model.hip expects.  H x W is the
// window; tr: model.hip holds windows with more columns than rows transposed (plane = [W+2][H+2]).
__global__ __launch_bounds__(256) void k_border_assemble(const float* __restrict__ q, const float* __restrict__ med,
                                                          const float* __restrict__ s1q, const float* __restrict__ s1med,
                                                          const float* __restrict__ dem, const float* __restrict__ aff, BTable bt,
                                                          Norm17 nm, int H, int W, int tr, long npix, int half,
                                                          float* __restrict__ frames, int* __restrict__ nonzero) {
#pragma clang fp contract(off)
    const int f = blockIdx.y, w = blockIdx.z;
    const int Wp = (tr ? H : W) + 2, Hp = (tr ? W : H) + 2, PP = (H + 2) * (W + 2);
    int p, py, px;
    if (tr) {
        // one 8 x 8 patch of the plane per wave: the source is read as 8 runs of 8 neighbouring pixels (448 B each), the
        // planes are written as 8 runs of 32 B -- a plain linear mapping would gather every source pixel from another row
        const int npx = (Wp + 7) >> 3, gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), l = threadIdx.x & 63;
        py = (gw / npx) * 8 + (l >> 3); px = (gw % npx) * 8 + (l & 7);
        p = (py < Hp && px < Wp) ? py * Wp + px : PP;
    } else {
        p = blockIdx.x * blockDim.x + threadIdx.x;
        py = p / Wp; px = p - py * Wp;
    }
    bool nz = false;
    if (p < PP) {
        const int wy = (tr ? px : py) - 1, wx = (tr ? py : px) - 1;
        const bool border = wy < 0 || wy >= H || wx < 0 || wx >= W;
        const BWin bw = bt.w[w];
        const int row = bw.start + reflect_idx(reflect_idx(wy, H) - bw.pad0, bw.h);
        const int col = reflect_idx(wx, W);
        const long tp = (long)row * W + col;
        const bool last = f == 4;
        const float* a = aff ? aff + (((long)(w * 5 + f) * 2 + (col >= half ? 1 : 0)) * 14) * 2 : nullptr;
        const float* s2 = last ? med + tp * 14 : q + ((long)f * npix + tp) * 14;
        const float* s1 = last ? s1med + tp * 2 : s1q + ((long)f * npix + tp) * 2;
        float* dst = frames + (((long)w * 5 + f) * 17) * PP + p;
        for (int c = 0; c < 17; ++c) {
            float v;
            if (c == 10) v = dem[tp];
            else if (c == 11 || c == 12) v = s1[c - 11];
            else {
                const int ch = c < 10 ? c : c - 3;
                v = s2[ch];
                if (a) v = v * a[2 * ch] + a[2 * ch + 1];
            }
            if (!border && v != 0.0f) nz = true;
            v = fminf(fmaxf(v, nm.lo[c]), nm.hi[c]);
            dst[(long)c * PP] = (last && border) ? 0.0f : (v - nm.mid[c]) / nm.half[c];
        }
    }
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(&nonzero[w], 1);
}

// probs [n][oh][ow] in place.  stats [n][4] = (max, mean, adjusted, filled)
__global__ __launch_bounds__(1024) void k_seam_adjust(float* __restrict__ probs, const int* __restrict__ nonzero, int dates_ok,
                                                       int oh, int ow, float* __restrict__ stats) {
#pragma clang fp contract(off)
    __shared__ double red[16];
    __shared__ float mxs[16];
    const int w = blockIdx.x, P = oh * ow, S = ow;
    float* pr = probs + (long)w * P;
    if (!nonzero[w] || !dates_ok) {
        for (int i = threadIdx.x; i < P; i += blockDim.x) pr[i] = 255.0f;
        if (threadIdx.x == 0) { stats[4 * w] = 255.0f; stats[4 * w + 1] = 255.0f; stats[4 * w + 2] = 0.0f; stats[4 * w + 3] = 1.0f; }
        return;
    }
    double sl = 0, sr = 0;
    for (int i = threadIdx.x; i < oh * 4; i += blockDim.x) {
        const int r = i >> 2, c = i & 3;
        sl += (double)pr[r * S + (S - 8) / 2 + c];
        sr += (double)pr[r * S + S / 2 + c];
    }
    sl = block_sum(sl, red); sr = block_sum(sr, red);
    const float lm = (float)(sl / (oh * 4.0)), rm = (float)(sr / (oh * 4.0));
    const bool adjust = fabsf(lm - rm) > 0.15f;
    if (adjust) {
        double s0 = 0, n0 = 0, s1 = 0, n1 = 0;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            const int c = i % S;
            const float v = pr[i];
            if (v > 0.05f) { if (c < S / 2) { s0 += (double)v; n0 += 1.0; } else { s1 += (double)v; n1 += 1.0; } }
        }
        s0 = block_sum(s0, red); n0 = block_sum(n0, red); s1 = block_sum(s1, red); n1 = block_sum(n1, red);
        const float adj = ((float)(s1 / n1) - (float)(s0 / n0)) / 2.0f;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            const int c = i % S;
            float v = pr[i];
            if (v > 0.05f) v = c < S / 2 ? v + adj : v - adj;
            pr[i] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
        __syncthreads();
    }
    double sum = 0; float mx = -INFINITY;
    for (int i = threadIdx.x; i < P; i += blockDim.x) { const float v = pr[i]; sum += (double)v; mx = fmaxf(mx, v); }
    sum = block_sum(sum, red);
    for (int k = 32; k >= 1; k >>= 1) mx = fmaxf(mx, __shfl_xor(mx, k));
    if ((threadIdx.x & 63) == 0) mxs[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, mxs[i]);
        stats[4 * w] = mx; stats[4 * w + 1] = (float)(sum / P); stats[4 * w + 2] = adjust ? 1.0f : 0.0f; stats[4 * w + 3] = 0.0f;
    }
}

// ---- border-aware mosaic: recreate_resegmented_tifs / mosaic_subtiles (:1169-1549) ----------------------------------
// One thread per output pixel walks the window table.  A window's prediction is used transposed and scaled by 100
// (:1300); border windows contribute the half that lies inside this tile (:1349-1351 and siblings).
constexpr int kMaxRW = 128;
struct RWin { int kind, x0, y0, sx, sy, ox, oy, cols; long pred_off, wt_off; };
struct RTable { int n; RWin w[kMaxRW]; };

// valid[w] = (sum of the saved window < sx * sy * 255), i.e. not the 255 fill (:1299, :1349, ...)
__global__ __launch_bounds__(256) void k_rwin_valid(const float* __restrict__ preds, RTable rt, const int* __restrict__ rows,
                                                     int* __restrict__ valid) {
    __shared__ double red[4];
    const RWin w = rt.w[blockIdx.x];
    const long n = (long)rows[blockIdx.x] * w.cols;
    double s = 0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += (double)preds[w.pred_off + i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) valid[blockIdx.x] = s < (double)w.sx * w.sy * 255.0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_reseg_mosaic(const float* __restrict__ preds, const float* __restrict__ wts, RTable rt,
                                                       const int* __restrict__ valid, const double* __restrict__ ramps,
                                                       unsigned kinds_present, int X, int Y, float* __restrict__ out,
                                                       float* __restrict__ sums_out) {
#pragma clang fp contract(off)
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= X * Y) return;
    const int x = id / Y, y = id - x * Y;
    int cnt_n = 0, cnt_n_na = 0, cnt_r = 0, cnt_r_na = 0;
    float Wk[5] = {0, 0, 0, 0, 0};
    bool has[5] = {false, false, false, false, false};
    for (int i = 0; i < rt.n; ++i) {
        const RWin& w = rt.w[i];
        const int lx = x - w.x0, ly = y - w.y0;
        if (!valid[i] || lx < 0 || ly < 0 || lx >= w.sx || ly >= w.sy) continue;
        const float p = preds[w.pred_off + (long)(ly + w.oy) * w.cols + lx + w.ox] * 100.0f;
        const bool na = p > 100.0f;
        if (w.kind == 0) { cnt_n++; cnt_n_na += na; } else { cnt_r++; cnt_r_na += na; }
        has[w.kind] = true;
        if (!na) Wk[w.kind] += wts[w.wt_off + (long)lx * w.sy + ly];
    }
    const double ramp_n = ramps[id];
    const bool nodata = (cnt_r == 0 && cnt_n - cnt_n_na == 0) || (cnt_r > 0 && (cnt_n - cnt_n_na == 0 || cnt_r_na > 0));
    if (nodata) { out[id] = 255.0f; if (sums_out) sums_out[id] = (float)ramp_n; return; }
    float Sk[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < rt.n; ++i) {
        const RWin& w = rt.w[i];
        const int lx = x - w.x0, ly = y - w.y0;
        if (!valid[i] || lx < 0 || ly < 0 || lx >= w.sx || ly >= w.sy) continue;
        const float p = preds[w.pred_off + (long)(ly + w.oy) * w.cols + lx + w.ox] * 100.0f;
        const float m = p > 100.0f ? 0.0f : wts[w.wt_off + (long)lx * w.sy + ly];
        const float t = p * (m / Wk[w.kind]);
        if (t == t) Sk[w.kind] += t;                           // nansum
    }
    const long XY = (long)X * Y;
    double mk[5];
    mk[0] = ramp_n;
    for (int k = 1; k < 5; ++k) mk[k] = (has[k] && ((kinds_present >> k) & 1u)) ? ramps[k * XY + id] : 0.0;
    const double sums = (((mk[1] + mk[2]) + mk[3]) + mk[4]) + mk[0];          // l + r + u + d + n (:1540)
    double v = (double)Sk[1] * (mk[1] / sums) + (double)Sk[4] * (mk[4] / sums);
    v = v + ((double)Sk[2] * (mk[2] / sums) + (double)Sk[3] * (mk[3] / sums));
    v = v + (double)Sk[0] * (mk[0] / sums);
    out[id] = v == v ? (float)v : 255.0f;
    if (sums_out) sums_out[id] = (float)sums;
}

}  // namespace

ttc_status reseg_mosaic(ttc_ctx* c, const float* d_preds, const ttc_reseg_window* h_wins, int n, const float* d_weights,
                        const double* d_ramps, int X, int Y, float* d_out, float* d_sums, hipStream_t s) {
    if (!d_preds || !h_wins || !d_weights || !d_ramps || !d_out || X < 1 || Y < 1) return c->fail(TTC_ERR_ARG, "reseg_mosaic: bad argument");
    if (n < 1 || n > kMaxRW) return c->fail(TTC_ERR_ARG, "reseg_mosaic: 1..128 windows");
    RTable rt{};
    rt.n = n;
    int h_rows[kMaxRW];
    unsigned present = 0;
    for (int i = 0; i < n; ++i) {
        const ttc_reseg_window& w = h_wins[i];
        if (w.kind < 0 || w.kind > 4 || w.rows < 2 || w.cols < 2) return c->fail(TTC_ERR_ARG, "reseg_mosaic: bad window");
        RWin r{};
        r.kind = w.kind; r.x0 = w.x; r.y0 = w.y; r.cols = w.cols; r.pred_off = w.pred_off; r.wt_off = w.weight_off;
        r.sx = w.cols; r.sy = w.rows; r.ox = 0; r.oy = 0;
        if (w.kind == 1 || w.kind == 2) { if (w.cols & 1) return c->fail(TTC_ERR_ARG, "reseg_mosaic: left / right windows need an even width"); r.sx = w.cols / 2; }
        if (w.kind == 3 || w.kind == 4) { if (w.rows & 1) return c->fail(TTC_ERR_ARG, "reseg_mosaic: up / down windows need an even height"); r.sy = w.rows / 2; }
        if (w.kind == 1) r.ox = r.sx;
        if (w.kind == 3) r.oy = r.sy;
        if (w.kind == 0 && r.sx != r.sy) return c->fail(TTC_ERR_ARG, "reseg_mosaic: plain windows are square");
        if (r.x0 < 0 || r.y0 < 0 || r.x0 + r.sx > X || r.y0 + r.sy > Y) return c->fail(TTC_ERR_ARG, "reseg_mosaic: window outside the tile");
        rt.w[i] = r; h_rows[i] = w.rows; present |= 1u << w.kind;
    }
    int* dv = static_cast<int*>(c->scratch_buf("rm_valid", sizeof(int) * 2 * kMaxRW));
    if (!dv) return c->fail(TTC_ERR_NOMEM, "reseg_mosaic scratch");
    int* drows = dv + kMaxRW;
    TTC_HIP(c, hipMemcpyAsync(drows, h_rows, sizeof(int) * n, hipMemcpyHostToDevice, s));
    TTC_HIP(c, hipStreamSynchronize(s));       // h_rows lives on this stack frame
    KTimer kt(c, "reseg_mosaic", s);
    hipLaunchKernelGGL(k_rwin_valid, dim3(n), dim3(256), 0, s, d_preds, rt, drows, dv);
    hipLaunchKernelGGL(k_reseg_mosaic, dim3((unsigned)(((long)X * Y + 255) / 256)), dim3(256), 0, s, d_preds, d_weights, rt, dv,
                       d_ramps, present, X, Y, d_out, d_sums);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status reseg_border_subtiles(ttc_ctx* c, const float* d_s2, const float* d_s1, const float* d_dem, int X,
                                 const int32_t* h_rows, int n, const float* h_min, const float* h_max, int hist_align,
                                 int n_dates_ok, float* d_preds, float* h_stats, int32_t* h_applied, hipStream_t s) {
    if (!d_s2 || !d_s1 || !d_dem || !h_rows || !h_min || !h_max || !d_preds || !h_stats)
        return c->fail(TTC_ERR_ARG, "border_subtiles: null argument");
    const int W = c->cfg.win_in, H = c->cfg.win_rows > 0 ? c->cfg.win_rows : c->cfg.win_in;
    if (c->cfg.length != 4) return c->fail(TTC_ERR_ARG, "border_subtiles: the border graph is quarterly (length 4)");
    if (n < 1 || n > kMaxBW || n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "border_subtiles: window count exceeds max_windows (or 8)");
    BTable bt{};
    bt.n = n;
    for (int i = 0; i < n; ++i) {
        const int start = h_rows[4 * i], h = h_rows[4 * i + 1], pad0 = h_rows[4 * i + 2], pad1 = h_rows[4 * i + 3];
        if (start < 0 || h < 8 || start + h > X || pad0 < 0 || pad1 < 0 || pad0 + h + pad1 != H || pad0 >= h || pad1 >= h)
            return c->fail(TTC_ERR_ARG, "border_subtiles: window rows do not fit the strip / win_rows");
        bt.w[i] = {start, h, pad0};
    }
    const long npix = (long)X * W;
    const int half = W / 2, oh = H - 14, ow = W - 14;
    float* q = static_cast<float*>(c->scratch_buf("bs_q", sizeof(float) * 4 * 14 * npix));
    float* med = static_cast<float*>(c->scratch_buf("bs_med", sizeof(float) * 14 * npix));
    float* s1q = static_cast<float*>(c->scratch_buf("bs_s1q", sizeof(float) * 4 * 2 * npix));
    float* s1med = static_cast<float*>(c->scratch_buf("bs_s1med", sizeof(float) * 2 * npix));
    constexpr int kHistChunks = 16;     // row chunks per (window, frame, half): 40 workgroups alone would leave the chip idle
    double* st = static_cast<double*>(c->scratch_buf("bs_stats", sizeof(double) * kMaxBW * 5 * 2 * kHistChunks * 42));
    float* aff = static_cast<float*>(c->scratch_buf("bs_aff", sizeof(float) * kMaxBW * 5 * 2 * 14 * 2));
    int* flags = static_cast<int*>(c->scratch_buf("bs_flags", sizeof(int) * (kMaxBW + kMaxBW * 5)));
    float* dstats = static_cast<float*>(c->scratch_buf("bs_dstats", sizeof(float) * kMaxBW * 4));
    if (!q || !med || !s1q || !s1med || !st || !aff || !flags || !dstats) return c->fail(TTC_ERR_NOMEM, "border scratch");
    c->named["border_q"] = {q, (size_t)4 * 14 * npix};
    c->named["border_med"] = {med, (size_t)14 * npix};
    int* applied = flags + kMaxBW;
    TTC_HIP(c, hipMemsetAsync(flags, 0, sizeof(int) * (kMaxBW + kMaxBW * 5), s));
    { KTimer kt(c, "border_medians", s);
      const long n2 = npix * 14, n1 = npix * 2;
      hipLaunchKernelGGL(k_steps_medians, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, d_s2, n2, 1, q, med);
      hipLaunchKernelGGL(k_steps_medians, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, d_s1, n1, 0, s1q, s1med);
      TTC_HIP(c, hipGetLastError()); }
    if (hist_align) {
        KTimer kt(c, "border_hist_align", s);
        hipLaunchKernelGGL(k_hist_stats, dim3(n * 5, 2, kHistChunks), dim3(256), 0, s, q, med, bt, W, npix, half, st);
        hipLaunchKernelGGL(k_hist_decide, dim3(n * 5), dim3(256), 0, s, q, med, bt, W, npix, half, (W - 14) / 2 + 7, st, kHistChunks,
                           aff, applied);
        TTC_HIP(c, hipGetLastError());
    }
    Norm17 nm{};
    for (int i = 0; i < 17; ++i) {
        // float32 throughout, as resegment_tiles_wide.py:1679-1685 builds them
        nm.lo[i] = h_min[i]; nm.hi[i] = h_max[i];
        nm.mid[i] = (h_max[i] + h_min[i]) / 2.0f;
        nm.half[i] = (h_max[i] - h_min[i]) / 2.0f;
    }
    { KTimer kt(c, "border_assemble", s);
      const int PP = (H + 2) * (W + 2), tr = H < W ? 1 : 0;
      const int patches = ((H + 2 + 7) / 8) * ((W + 2 + 7) / 8);              // 8 x 8 patches, one per wave (transposed planes)
      const int blocks = tr ? (patches + 3) / 4 : (PP + 255) / 256;
      c->frames_planar_valid = true;
      hipLaunchKernelGGL(k_border_assemble, dim3(blocks, 5, n), dim3(256), 0, s, q, med, s1q, s1med, d_dem,
                         hist_align ? aff : nullptr, bt, nm, H, W, tr, npix, half, c->frames, flags);
      TTC_HIP(c, hipGetLastError()); }
    TTC_CHECK(model_forward_frames(c, n, d_preds, s, FRAMES_PLANAR));
    { KTimer kt(c, "border_seam_adjust", s);
      hipLaunchKernelGGL(k_seam_adjust, dim3(n), dim3(1024), 0, s, d_preds, flags, n_dates_ok >= 2 ? 1 : 0, oh, ow, dstats);
      TTC_HIP(c, hipGetLastError()); }
    TTC_HIP(c, hipMemcpyAsync(h_stats, dstats, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, s));
    if (h_applied) TTC_HIP(c, hipMemcpyAsync(h_applied, applied, sizeof(int) * 5 * n, hipMemcpyDeviceToHost, s));
    TTC_HIP(c, hipStreamSynchronize(s));
    return TTC_OK;
}

ttc_status reseg_seam_adjust(ttc_ctx* c, float* d_preds, int n, int oh, int ow, float* h_stats, hipStream_t s) {
    if (!d_preds || !h_stats || n < 1 || oh < 1 || ow < 16) return c->fail(TTC_ERR_ARG, "seam_adjust: bad argument");
    int* flags = static_cast<int*>(c->scratch_buf("sa_flags", sizeof(int) * n));
    float* dstats = static_cast<float*>(c->scratch_buf("sa_dstats", sizeof(float) * 4 * n));
    if (!flags || !dstats) return c->fail(TTC_ERR_NOMEM, "seam_adjust scratch");
    TTC_HIP(c, hipMemsetAsync(flags, 1, sizeof(int) * n, s));
    { KTimer kt(c, "border_seam_adjust", s);
      hipLaunchKernelGGL(k_seam_adjust, dim3(n), dim3(1024), 0, s, d_preds, flags, 1, oh, ow, dstats);
      TTC_HIP(c, hipGetLastError()); }
    TTC_HIP(c, hipMemcpyAsync(h_stats, dstats, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, s));
    TTC_HIP(c, hipStreamSynchronize(s));
    return TTC_OK;
}
