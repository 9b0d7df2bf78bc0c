// ConvGRU + U-Net forward over a batch of windows: elementwise / normalisation kernels and
// the launch sequence around the MFMA conv engine (conv3x3_mfma.hip).
//
// Reference semantics: src/train/src/model.py (group_norm :100-121, ConvGRUCell :240-290,
// ZoneoutWrapper :556-579, conv_swish_gn :448-538, sse_block :45-61) assembled as in
// src/train/train-model.py:140-231.  The reference runs batch 1 per Session.run
// (job.py:353-357); here all windows of one or more tiles are one batch, and the fw / bw
// ConvGRU directions are a second batch axis (sequence n' = dir*N + n).
//
// Activation layout: planar fp32, [n][C][H][W]; every tensor that feeds a 3x3 conv is
// stored already padded (reflect for the ConvGRU, zero for SAME blocks) so the conv is a
// pure linear-offset implicit GEMM.  GroupNorm needs whole-window statistics (SURVEY.md
// F7): the conv epilogue emits per-workgroup partial sums, k_gn_finalize reduces them in
// double, and the NEXT elementwise kernel applies the affine while it gathers
// (pad / pool / upsample / crop / concat are index math, never separate passes).
#include <algorithm>

#include "h16_common.h"

namespace {

using B16 = ttc_ctx::B16;

// Activations on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute error <= 2e-7, two
// orders below the fp32 summation-order differences the parity tests already allow; libm's expf / tanhf / IEEE division made
// the GRU update kernels VALU-bound instead of HBM-bound.
__device__ __forceinline__ float sigm(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * v)); }
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ---------------------------------------------------------------------------------------
// [n][L+1][rows][cols][C] (reference feed layout) -> padded planar frames [n][L+1][C][H+2][W+2];
// frames < L reflect-padded (ConvGRU, model.py:250), frame L zero-padded (SAME conv).
// H, W are the INTERNAL plane dims; tr: the planes hold the window transposed (rows = W, cols = H), see Geo.
__global__ void k_nhwc_to_frames(const float* __restrict__ in, float* __restrict__ frames, int L1, int H, int W, int C, int tr) {
    const int Wp = W + 2, PP = (H + 2) * Wp;
    const int f = blockIdx.y, n = blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, px = p - py * Wp;
    const bool last = (f == L1 - 1);
    int sy = py - 1, sx = px - 1;
    const bool border = sy < 0 || sy >= H || sx < 0 || sx >= W;
    sy = reflect_idx(sy, H); sx = reflect_idx(sx, W);
    const float* src = in + (tr ? (((long)n * L1 + f) * W + sx) * H + sy : (((long)n * L1 + f) * H + sy) * W + sx) * C;
    float* dst = frames + (((long)n * L1 + f) * C) * PP + p;
    for (int c = 0; c < C; ++c) dst[(long)c * PP] = (last && border) ? 0.0f : src[c];
}

// ---------------------------------------------------------------------------------------
// GN partial sums -> (mean, rstd).  stats: [n][Cout/4][nblk = tiles * waves][2]; groups of `qpg` quads.
__global__ void k_gn_finalize(const float* __restrict__ stats, float* __restrict__ gn, int nquads, int nblk,
                              int qpg, double count, float eps) {
    const int g = blockIdx.x, n = blockIdx.y, G = gridDim.x;
    const float* src = stats + (((long)n * nquads + (long)g * qpg) * nblk) * 2;
    double s = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < qpg * nblk; i += 64) { s += src[2 * i]; s2 += src[2 * i + 1]; }
    for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m); s2 += __shfl_xor(s2, m); }
    if (threadIdx.x == 0) {
        const double mean = s / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        gn[((long)n * G + g) * 2 + 0] = (float)mean;
        gn[((long)n * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// the first `cnt` floats of every sequence n of two tensors with `stride_n` floats per sequence := 0 (hipMemset2DAsync measured
// 3 ms per call on this shape)
__global__ void k_zero_planes(float* __restrict__ a, float* __restrict__ b, long stride_n, long cnt) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    a[(long)blockIdx.y * stride_n + i] = 0.0f;
    b[(long)blockIdx.y * stride_n + i] = 0.0f;
}

// ---------------------------------------------------------------------------------------
// ConvGRU, after the gates conv (model.py:259-270):
//   r = sigmoid(GN(y[:32])); writes r*h (reflect-padded).  The update gate u = sigmoid(GN(y[32:])) is formed where it
//   is consumed (k_gru_apply2) instead of making a round trip through HBM.
struct GruParams { const float* base; long dir_stride; };   // per-direction parameter block
// parameter block layout (floats): gr[32] br[32] gu[32] bu[32] k1[32] gy[32] by[32]

__global__ void k_gru_apply1(const float* __restrict__ yg, const float* __restrict__ gn, GruParams prm,
                             const float* __restrict__ hcur, float* __restrict__ rh, int H, int W, int N) {
    const int Wp = W + 2, PP = (H + 2) * Wp, P = H * Wp;        // raw conv outputs keep the input pitch: [c][H][Wp]
    const int n = blockIdx.y, dir = n / N;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, px = p - py * Wp;
    const int s = reflect_idx(py - 1, H) * Wp + reflect_idx(px - 1, W);
    const float* pr = prm.base + dir * prm.dir_stride;
    const float* y = yg + (long)n * 64 * P + s;
    const float* g = gn + (long)n * 32;      // 16 groups x (mean, rstd): 0-7 r, 8-15 u
    const float* h = hcur + (long)n * 32 * PP + p;
    float* o = rh + (long)n * 32 * PP + p;
#pragma unroll 4
    for (int c = 0; c < 32; ++c) {
        const int gi = c >> 2;
        const float r = sigm((y[(long)c * P] - g[2 * gi]) * g[2 * gi + 1] * pr[c] + pr[32 + c]);
        o[(long)c * PP] = r * h[(long)c * PP];
    }
}

// after the candidate conv (its epilogue already applied the in-cell sSE, model.py:278-282):
//   cand = tanh(GN(y)); h' = u*h + (1-u)*cand (model.py:288); carried state = z*h + (1-z)*h'
//   (ZoneoutWrapper inference branch, model.py:571-574).  On the last step also writes the
//   final state zero-padded into gru_out[n][dir*32 + c] (train-model.py:165 concat order).
__global__ void k_gru_apply2(const float* __restrict__ yc, const float* __restrict__ gn, GruParams prm,
                             const float* __restrict__ yg, const float* __restrict__ gn_gates, float* __restrict__ u_keep,
                             const float* __restrict__ hcur, float* __restrict__ hnext,
                             float* __restrict__ gru_out, int H, int W, int N, float z, int h_zero) {
    // h_zero: the incoming state is identically zero (step 0) and is NOT read (its buffer may hold a previous tile's state)
    const int Wp = W + 2, PP = (H + 2) * Wp, P = H * Wp;        // raw conv outputs keep the input pitch: [c][H][Wp]
    const int n = blockIdx.y, dir = n / N;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, px = p - py * Wp;
    const int sy0 = py - 1, sx0 = px - 1;
    const bool interior = sy0 >= 0 && sy0 < H && sx0 >= 0 && sx0 < W;
    const int s = reflect_idx(sy0, H) * Wp + reflect_idx(sx0, W);
    const int su = reflect_idx(sy0, H) * W + reflect_idx(sx0, W);      // the debug copy of u stays [c][H][W]
    const float* pr = prm.base + dir * prm.dir_stride;
    const float* y = yc + (long)n * 32 * P + s;
    const float* g = gn + (long)n * 16;      // 8 groups x (mean, rstd)
    const float* yu = yg + ((long)n * 64 + 32) * P + s;
    const float* gu = gn_gates + (long)n * 32 + 16;   // gates groups 8-15 = u
    const float* h = hcur + (long)n * 32 * PP + p;
    float* hn = hnext ? hnext + (long)n * 32 * PP + p : nullptr;     // nullptr on the last step: nobody reads the state afterwards
    float* go = gru_out ? gru_out + ((long)(n - dir * N) * 64 + dir * 32) * PP + p : nullptr;
    float* uk = (u_keep && interior) ? u_keep + (long)n * 32 * (H * W) + su : nullptr;
#pragma unroll 4
    for (int c = 0; c < 32; ++c) {
        const int gi = c >> 2;
        const float cand = tanh_fast((y[(long)c * P] - g[2 * gi]) * g[2 * gi + 1] * pr[160 + c] + pr[192 + c]);
        const float uv = sigm((yu[(long)c * P] - gu[2 * gi]) * gu[2 * gi + 1] * pr[64 + c] + pr[96 + c]);
        if (uk) uk[(long)c * (H * W)] = uv;
        const float hv = h_zero ? 0.0f : h[(long)c * PP];
        const float hnew = uv * hv + (1.0f - uv) * cand;
        const float hz = hv * z + hnew * (1.0f - z);
        if (hn) hn[(long)c * PP] = hz;
        if (go) go[(long)c * PP] = interior ? hz : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------
// conv_swish_gn tail (model.py:519-531): GN(8 groups) affine -> sSE gate x * sigmoid(w.x + b),
// fused with the gather that builds the next conv's input.
enum GatherMode : int { G_COPY = 0, G_POOL = 1, G_UP = 2 };
// block parameter layout (floats): gamma[C] beta[C] ssew[C] sseb[1]
struct FinArgs {
    const float* y; const float* gn; const float* prm;
    float* dst;
    int C, Hs, Ws;        // source (raw conv output) dims; its rows have pitch Ws + 2 (the conv's input pitch)
    int Hd, Wd;           // destination dims INCLUDING pad
    int pad, crop, mode;
    long dst_stride_n;    // floats per n in dst (lets two producers share a concat buffer)
    int dst_coff;         // channel offset inside dst
    // 16-bit engine: channel-blocked destination instead of dst; dst_stride_n then counts 16-byte units and dst_coff blocks
    uint4* dhi = nullptr; uint4* dlo = nullptr;
    int nraw = 0;         // 16-bit engine: the n the PRODUCING conv wrote the raw split planes with (the bottom plane starts after nraw * (C / 8) * PS
                          // units) -- stated by the launcher, never derived from this kernel's own grid
};

template <int MODE>
__global__ void k_block_finalize(FinArgs a) {
    extern __shared__ float sm[];            // scale[C] shift[C] ssew[C]
    const int C = a.C, n = blockIdx.y;
    const int cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = a.gn[((long)n * 8 + c / cpg) * 2], rstd = a.gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * a.prm[c];
        sm[c] = sc; sm[C + c] = a.prm[C + c] - mean * sc; sm[2 * C + c] = a.prm[2 * C + c];
    }
    __syncthreads();
    const float sseb = a.prm[3 * C];
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.Hd * a.Wd) return;
    const int dy = d / a.Wd, dx = d - dy * a.Wd;
    const int SP = a.Ws + 2;
    const long PS = (long)a.Hs * SP, PD = (long)a.Hd * a.Wd;
    float* dst = a.dst + (long)n * a.dst_stride_n + (long)a.dst_coff * PD + d;
    const int iy = dy - a.pad, ix = dx - a.pad;
    if (iy < 0 || ix < 0 || iy >= a.Hd - 2 * a.pad || ix >= a.Wd - 2 * a.pad) {
        for (int c = 0; c < C; ++c) dst[(long)c * PD] = 0.0f;
        return;
    }
    constexpr int NS = (MODE == G_POOL) ? 4 : 1;
    int src[NS];
    if (MODE == G_COPY) src[0] = (iy + a.crop) * SP + ix + a.crop;
    if (MODE == G_UP) src[0] = (iy >> 1) * SP + (ix >> 1);
    if (MODE == G_POOL) {
        src[0] = (2 * iy) * SP + 2 * ix; src[1] = src[0] + 1; src[2] = src[0] + SP; src[3] = src[2] + 1;
    }
    const float* y = a.y + (long)n * C * PS;
    float gate[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) gate[k] = sseb;
    for (int c = 0; c < C; ++c) {
        const float sc = sm[c], sh = sm[C + c], w = sm[2 * C + c];
#pragma unroll
        for (int k = 0; k < NS; ++k) gate[k] += w * (y[(long)c * PS + src[k]] * sc + sh);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) gate[k] = sigm(gate[k]);
    for (int c = 0; c < C; ++c) {
        const float sc = sm[c], sh = sm[C + c];
        float v = (y[(long)c * PS + src[0]] * sc + sh) * gate[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) v = fmaxf(v, (y[(long)c * PS + src[k]] * sc + sh) * gate[k]);
        dst[(long)c * PD] = v;
    }
}

// The same for the single-source gathers (COPY / UP) of the C = 64 and C = 128 blocks: the thread keeps its C normalised values
// in registers between the gate sum and the output pass, so the raw conv output is read from HBM ONCE (the two-pass form above
// read it twice: these passes are pure HBM streams, and the raw tensors do not fit the L2).
template <int MODE, int C>
__global__ __launch_bounds__(256) void k_block_finalize_1p(FinArgs a) {
    static_assert(MODE != G_POOL, "the pooled gather has four sources per destination");
    __shared__ float sm[3 * C];              // scale[C] shift[C] ssew[C]
    const int n = blockIdx.y;
    constexpr int cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = a.gn[((long)n * 8 + c / cpg) * 2], rstd = a.gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * a.prm[c];
        sm[c] = sc; sm[C + c] = a.prm[C + c] - mean * sc; sm[2 * C + c] = a.prm[2 * C + c];
    }
    __syncthreads();
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.Hd * a.Wd) return;
    const int dy = d / a.Wd, dx = d - dy * a.Wd;
    const int SP = a.Ws + 2;
    const long PS = (long)a.Hs * SP, PD = (long)a.Hd * a.Wd;
    float* dst = a.dst + (long)n * a.dst_stride_n + (long)a.dst_coff * PD + d;
    const int iy = dy - a.pad, ix = dx - a.pad;
    if (iy < 0 || ix < 0 || iy >= a.Hd - 2 * a.pad || ix >= a.Wd - 2 * a.pad) {
        for (int c = 0; c < C; ++c) dst[(long)c * PD] = 0.0f;
        return;
    }
    const int src = MODE == G_COPY ? (iy + a.crop) * SP + ix + a.crop : (iy >> 1) * SP + (ix >> 1);
    const float* y = a.y + (long)n * C * PS + src;
    float zn[C];
    float gate = a.prm[3 * C];
#pragma unroll
    for (int c = 0; c < C; ++c) { zn[c] = y[(long)c * PS] * sm[c] + sm[C + c]; gate += sm[2 * C + c] * zn[c]; }
    gate = sigm(gate);
#pragma unroll
    for (int c = 0; c < C; ++c) dst[(long)c * PD] = zn[c] * gate;
}

// final block tail + 1x1 head (train-model.py:226-231): sigmoid(sum_c hw[c] * z[c] + hb)
// BLK: the raw conv output is the 16-bit engine's channel-blocked split planes (Raw16, h16_common.h), `yraw` = its top plane
// and the bottom plane follows after nraw * (C / 8) * PS units (nraw = the n of the conv launch that wrote it)
template <bool BLK>
__global__ void k_head(const float* __restrict__ yraw, const float* __restrict__ gn, const float* __restrict__ prm,
                       const float* __restrict__ headp, float* __restrict__ out, int C, int P, int ow, int tr, int nraw) {
    extern __shared__ float sm[];            // scale shift ssew headw
    const int n = blockIdx.y, cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = gn[((long)n * 8 + c / cpg) * 2], rstd = gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * prm[c];
        sm[c] = sc; sm[C + c] = prm[C + c] - mean * sc; sm[2 * C + c] = prm[2 * C + c]; sm[3 * C + c] = headp[c];
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int oh = P / ow;
    const long PS = (long)oh * (ow + 2);                                   // raw plane with the conv's input pitch
    const long s0 = (p / ow) * (ow + 2) + p % ow;
    float gate = prm[3 * C];
    float logit = headp[C];
    if constexpr (BLK) {
        const uint4* top = reinterpret_cast<const uint4*>(yraw);
        const uint4* bot = top + (long)nraw * (C / 8) * PS;
        float z = 0.f;                                                     // sum_c hw[c] * zn[c]: the gate multiplies it afterwards
        for (int k = 0; k < C / 8; ++k) {
            float v[8];
            raw_load8(top, bot, ((long)n * (C / 8) + k) * PS + s0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * k + j;
                const float zn = v[j] * sm[c] + sm[C + c];
                gate += sm[2 * C + c] * zn;
                z += sm[3 * C + c] * zn;
            }
        }
        logit += z * sigm(gate);
    } else {
        const float* y = yraw + (long)n * C * PS + s0;
        for (int c = 0; c < C; ++c) gate += sm[2 * C + c] * (y[(long)c * PS] * sm[c] + sm[C + c]);
        gate = sigm(gate);
        for (int c = 0; c < C; ++c) logit += sm[3 * C + c] * ((y[(long)c * PS] * sm[c] + sm[C + c]) * gate);
    }
    out[(long)n * P + (tr ? (p % ow) * oh + p / ow : p)] = sigm(logit);      // tr: the plane is the transposed window
}

// ---- feature taps (--gen_feats, job.py:1429-1445, tensors named at :1808-1809) -------------------------
// early = the bi-ConvGRU output (`gru_drop/.../Merge:0`, inference: identity), [n, rows, cols, 64] NHWC
// (H, W internal plane dims; tr: rows = W, cols = H)
__global__ void k_tap_early(const float* __restrict__ gru_out, int H, int W, int N, int tr, float* __restrict__ out) {
    const int Wp = W + 2;
    const long PP = (long)(H + 2) * Wp, total = (long)N * H * W * 64;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int ch = (int)(id & 63);
    const long pix = id >> 6;
    const int n = (int)(pix / ((long)H * W)), r = (int)(pix - (long)n * H * W);
    const int cols = tr ? H : W;
    const int uy = r / cols, ux = r - uy * cols;
    const int y = tr ? ux : uy, x = tr ? uy : ux;
    out[id] = gru_out[((long)n * 64 + ch) * PP + (long)(y + 1) * Wp + (x + 1)];
}
// late = output of the last conv_swish_gn block after its sSE gate (`csse_out_mul/mul:0`), [n, o, o, C] NHWC
template <bool BLK>
__global__ void k_tap_late(const float* __restrict__ yraw, const float* __restrict__ gn, const float* __restrict__ prm,
                           float* __restrict__ out, int C, int P, int ow, int tr, int nraw) {
    extern __shared__ float sm[];            // scale shift ssew
    const int n = blockIdx.y, cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = gn[((long)n * 8 + c / cpg) * 2], rstd = gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * prm[c];
        sm[c] = sc; sm[C + c] = prm[C + c] - mean * sc; sm[2 * C + c] = prm[2 * C + c];
    }
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int oh = P / ow;
    const long PS = (long)oh * (ow + 2);                                   // raw plane with the conv's input pitch
    const long s0 = (p / ow) * (ow + 2) + p % ow;
    float gate = prm[3 * C];
    float* o = out + ((long)n * P + (tr ? (p % ow) * oh + p / ow : p)) * C;
    if constexpr (BLK) {
        const uint4* top = reinterpret_cast<const uint4*>(yraw);
        const uint4* bot = top + (long)nraw * (C / 8) * PS;
        for (int k = 0; k < C / 8; ++k) {
            float v[8];
            raw_load8(top, bot, ((long)n * (C / 8) + k) * PS + s0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) gate += sm[2 * C + 8 * k + j] * (v[j] * sm[8 * k + j] + sm[C + 8 * k + j]);
        }
        gate = sigm(gate);
        for (int k = 0; k < C / 8; ++k) {
            float v[8];
            raw_load8(top, bot, ((long)n * (C / 8) + k) * PS + s0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[8 * k + j] = (v[j] * sm[8 * k + j] + sm[C + 8 * k + j]) * gate;
        }
    } else {
        const float* y = yraw + (long)n * C * PS + s0;
        for (int c = 0; c < C; ++c) gate += sm[2 * C + c] * (y[(long)c * PS] * sm[c] + sm[C + c]);
        gate = sigm(gate);
        for (int c = 0; c < C; ++c) o[c] = (y[(long)c * PS] * sm[c] + sm[C + c]) * gate;
    }
}

// ======================================================================================================================
// 16-bit engine (cfg.precision >= 2, conv3x3_h16.hip): the same elementwise stages, reading the fp32 raw conv outputs and
// writing the NEXT conv's input channel-blocked as hi + lo 16-bit K vectors ([n][C8][plane][8]).  The ConvGRU state lives
// only as such a pair (fp16: 22 mantissa bits).
template <int BF>
__global__ void k_gru_apply1_b16(Raw16 yg, const float* __restrict__ gn, GruParams prm, B16 hcur, B16 rh,
                                 int H, int W, int N) {
    const int Wp = W + 2, PP = (H + 2) * Wp, P = H * Wp;        // raw conv outputs keep the input pitch: [c][H][Wp]
    const int n = blockIdx.y, dir = n / N;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, px = p - py * Wp;
    const int s = reflect_idx(py - 1, H) * Wp + reflect_idx(px - 1, W);
    const float* pr = prm.base + dir * prm.dir_stride;
    const float* g = gn + (long)n * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long u = ((long)n * 4 + k) * PP + p;
        float hv[8], o[8], y[8];
        b16_load8<BF>(hcur.hi, hcur.lo, u, hv);
        raw_load8(yg.top, yg.bot, ((long)n * 8 + k) * P + s, y);          // gates channels 0..31 = r
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * k + j, gi = c >> 2;
            const float r = sigm((y[j] - g[2 * gi]) * g[2 * gi + 1] * pr[c] + pr[32 + c]);
            o[j] = r * hv[j];
        }
        b16_store8<BF>(rh.hi, rh.lo, u, o);
    }
}

template <int BF>
__global__ void k_gru_apply2_b16(Raw16 yc, const float* __restrict__ gn, GruParams prm,
                                 Raw16 yg, const float* __restrict__ gn_gates, float* __restrict__ u_keep,
                                 B16 hcur, B16 hnext, B16 gru_out, int H, int W, int N, float z, int h_zero) {
    // h_zero: the incoming state is identically zero (step 0) and is NOT read
    const int Wp = W + 2, PP = (H + 2) * Wp, P = H * Wp;        // raw conv outputs keep the input pitch: [c][H][Wp]
    const int n = blockIdx.y, dir = n / N;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const int py = p / Wp, px = p - py * Wp;
    const int sy0 = py - 1, sx0 = px - 1;
    const bool interior = sy0 >= 0 && sy0 < H && sx0 >= 0 && sx0 < W;
    const int s = reflect_idx(sy0, H) * Wp + reflect_idx(sx0, W);
    const int su = reflect_idx(sy0, H) * W + reflect_idx(sx0, W);      // the debug copy of u stays [c][H][W]
    const float* pr = prm.base + dir * prm.dir_stride;
    const float* g = gn + (long)n * 16;
    const float* gu = gn_gates + (long)n * 32 + 16;
    float* uk = (u_keep && interior) ? u_keep + (long)n * 32 * (H * W) + su : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long u = ((long)n * 4 + k) * PP + p;
        float hv[8], o[8], y[8], yu[8];
        if (h_zero) {
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = 0.0f;
        } else b16_load8<BF>(hcur.hi, hcur.lo, u, hv);
        raw_load8(yc.top, yc.bot, ((long)n * 4 + k) * P + s, y);
        raw_load8(yg.top, yg.bot, ((long)n * 8 + 4 + k) * P + s, yu);      // gates channels 32..63 = u
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * k + j, gi = c >> 2;
            const float cand = tanh_fast((y[j] - g[2 * gi]) * g[2 * gi + 1] * pr[160 + c] + pr[192 + c]);
            const float uv = sigm((yu[j] - gu[2 * gi]) * gu[2 * gi + 1] * pr[64 + c] + pr[96 + c]);
            if (uk) uk[(long)c * (H * W)] = uv;
            const float hnew = uv * hv[j] + (1.0f - uv) * cand;
            o[j] = hv[j] * z + hnew * (1.0f - z);
        }
        if (hnext.hi) b16_store8<BF>(hnext.hi, hnext.lo, u, o);      // not on the last step: nobody reads the state afterwards
        if (gru_out.hi) {
            if (!interior) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = 0.0f;
            }
            b16_store8<BF>(gru_out.hi, gru_out.lo, ((long)(n - dir * N) * 8 + dir * 4 + k) * PP + p, o);
        }
    }
}

template <int BF, int MODE>
__global__ void k_block_finalize_b16(FinArgs a) {
    extern __shared__ float sm[];            // scale[C] shift[C] ssew[C]
    const int C = a.C, n = blockIdx.y;
    const int cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = a.gn[((long)n * 8 + c / cpg) * 2], rstd = a.gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * a.prm[c];
        sm[c] = sc; sm[C + c] = a.prm[C + c] - mean * sc; sm[2 * C + c] = a.prm[2 * C + c];
    }
    __syncthreads();
    const float sseb = a.prm[3 * C];
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.Hd * a.Wd) return;
    const int dy = d / a.Wd, dx = d - dy * a.Wd;
    const int SP = a.Ws + 2;
    const long PS = (long)a.Hs * SP, PD = (long)a.Hd * a.Wd;
    const long u0 = (long)n * a.dst_stride_n + (long)a.dst_coff * PD + d;
    const int iy = dy - a.pad, ix = dx - a.pad;
    if (iy < 0 || ix < 0 || iy >= a.Hd - 2 * a.pad || ix >= a.Wd - 2 * a.pad) {
        const uint4 zv = make_uint4(0, 0, 0, 0);
        for (int k = 0; k < C / 8; ++k) { a.dhi[u0 + k * PD] = zv; a.dlo[u0 + k * PD] = zv; }
        return;
    }
    constexpr int NS = (MODE == G_POOL) ? 4 : 1;
    int src[NS];
    if (MODE == G_COPY) src[0] = (iy + a.crop) * SP + ix + a.crop;
    if (MODE == G_UP) src[0] = (iy >> 1) * SP + (ix >> 1);
    if (MODE == G_POOL) {
        src[0] = (2 * iy) * SP + 2 * ix; src[1] = src[0] + 1; src[2] = src[0] + SP; src[3] = src[2] + 1;
    }
    // raw conv output: the 16-bit engine's channel-blocked split planes (a.y = top plane, the bottom plane follows)
    const uint4* ytop = reinterpret_cast<const uint4*>(a.y);
    const uint4* ybot = ytop + (long)a.nraw * (C / 8) * PS;
    const long ub = (long)n * (C / 8) * PS;
    float gate[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) gate[k] = sseb;
    for (int cb = 0; cb < C / 8; ++cb) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float v[8];
            raw_load8(ytop, ybot, ub + (long)cb * PS + src[k], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) gate[k] += sm[2 * C + 8 * cb + j] * (v[j] * sm[8 * cb + j] + sm[C + 8 * cb + j]);
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) gate[k] = sigm(gate[k]);
    for (int cb = 0; cb < C / 8; ++cb) {
        float o[8];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float v[8];
            raw_load8(ytop, ybot, ub + (long)cb * PS + src[k], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = (v[j] * sm[8 * cb + j] + sm[C + 8 * cb + j]) * gate[k];
                o[j] = k == 0 ? t : fmaxf(o[j], t);
            }
        }
        b16_store8<BF>(a.dhi, a.dlo, u0 + cb * PD, o);
    }
}

// one-pass form for the single-source gathers of the C = 64 / 128 blocks (see k_block_finalize_1p)
template <int BF, int MODE, int C>
__global__ __launch_bounds__(256) void k_block_finalize_b16_1p(FinArgs a) {
    static_assert(MODE != G_POOL, "the pooled gather has four sources per destination");
    __shared__ float sm[3 * C];
    const int n = blockIdx.y;
    constexpr int cpg = C / 8;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float mean = a.gn[((long)n * 8 + c / cpg) * 2], rstd = a.gn[((long)n * 8 + c / cpg) * 2 + 1];
        const float sc = rstd * a.prm[c];
        sm[c] = sc; sm[C + c] = a.prm[C + c] - mean * sc; sm[2 * C + c] = a.prm[2 * C + c];
    }
    __syncthreads();
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.Hd * a.Wd) return;
    const int dy = d / a.Wd, dx = d - dy * a.Wd;
    const int SP = a.Ws + 2;
    const long PS = (long)a.Hs * SP, PD = (long)a.Hd * a.Wd;
    const long u0 = (long)n * a.dst_stride_n + (long)a.dst_coff * PD + d;
    const int iy = dy - a.pad, ix = dx - a.pad;
    if (iy < 0 || ix < 0 || iy >= a.Hd - 2 * a.pad || ix >= a.Wd - 2 * a.pad) {
        const uint4 zv = make_uint4(0, 0, 0, 0);
        for (int k = 0; k < C / 8; ++k) { a.dhi[u0 + k * PD] = zv; a.dlo[u0 + k * PD] = zv; }
        return;
    }
    const int src = MODE == G_COPY ? (iy + a.crop) * SP + ix + a.crop : (iy >> 1) * SP + (ix >> 1);
    const uint4* ytop = reinterpret_cast<const uint4*>(a.y);
    const uint4* ybot = ytop + (long)a.nraw * (C / 8) * PS;
    const long ub = (long)n * (C / 8) * PS + src;
    float zn[C];
    float gate = a.prm[3 * C];
#pragma unroll
    for (int cb = 0; cb < C / 8; ++cb) {
        float v[8];
        raw_load8(ytop, ybot, ub + (long)cb * PS, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { zn[8 * cb + j] = v[j] * sm[8 * cb + j] + sm[C + 8 * cb + j]; gate += sm[2 * C + 8 * cb + j] * zn[8 * cb + j]; }
    }
    gate = sigm(gate);
#pragma unroll
    for (int cb = 0; cb < C / 8; ++cb) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = zn[8 * cb + j] * gate;
        b16_store8<BF>(a.dhi, a.dlo, u0 + cb * PD, o);
    }
}

template <int BF>
__global__ void k_tap_early_b16(B16 gru_out, int H, int W, int N, int tr, float* __restrict__ out) {
    const int Wp = W + 2;
    const long PP = (long)(H + 2) * Wp, total = (long)N * H * W * 8;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one channel block of one pixel
    if (id >= total) return;
    const int blk = (int)(id & 7);
    const long pix = id >> 3;
    const int n = (int)(pix / ((long)H * W)), r = (int)(pix - (long)n * H * W);
    const int cols = tr ? H : W;
    const int uy = r / cols, ux = r - uy * cols;
    const int y = tr ? ux : uy, x = tr ? uy : ux;
    float v[8];
    b16_load8<BF>(gru_out.hi, gru_out.lo, ((long)n * 8 + blk) * PP + (long)(y + 1) * Wp + (x + 1), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[pix * 64 + blk * 8 + j] = v[j];
}

// U-Net geometry of one axis (train-model.py:140-231): pool, VALID conv, pool, VALID conv, x2, x2, VALID conv
struct Axis {
    int n, np, c1, c2, u2, u3, o;
    explicit Axis(int n_) : n(n_), np(n_ + 2) { c1 = n / 2 - 2; c2 = c1 / 2 - 2; u2 = 2 * c2; u3 = 2 * u2; o = u3 - 2; }
};
// The conv engine tiles a FLATTENED padded plane: a 512-position tile stages 512 + 2 * (cols + 2) + 2 positions per input
// channel, so wide planes waste LDS and loads (686 columns: 1886 staged per 512 computed, one workgroup per CU).  A window
// with more columns than rows (the 220 x 684 border graph) is therefore held TRANSPOSED: planes are [cols][rows], every
// 3x3 kernel is uploaded with its taps swapped, and only the kernels at the boundary (frames in, probabilities / taps out,
// reseg.hip's assembly) know.  Everything else sees `y` = plane rows, `x` = plane columns.
inline int model_rows(const ttc_config& c) { return c.win_rows > 0 ? c.win_rows : c.win_in; }
inline bool model_transposed(const ttc_config& c) { return model_rows(c) < c.win_in; }
struct Geo {
    bool tr;
    Axis y, x;      // plane rows / columns
    int L;
    explicit Geo(const ttc_config& c)
        : tr(model_transposed(c)), y(tr ? c.win_in : model_rows(c)), x(tr ? model_rows(c) : c.win_in), L(c.length) {}
};

const char* kBlockNames[8] = {"conv_median", "conv_concat", "conv1", "conv2", "up2", "up2_out", "up3", "out"};
const int kBlockCin[8] = {17, 128, 64, 128, 256, 256, 128, 128};
const int kBlockCout[8] = {64, 64, 128, 256, 128, 128, 64, 64};

}  // namespace

size_t ttc_ctx::guard_bytes() {
    static const size_t g = [] { const char* e = getenv("TTC_GUARD"); const long k = e ? atol(e) : 0; return k > 0 ? (size_t)k * 1024 : (size_t)0; }();
    return g;
}

// hipMalloc with (TTC_GUARD) a 0xA5-filled zone before and after the buffer; the user pointer keeps hipMalloc's 4-KiB alignment
void* ttc_ctx::guarded_malloc(size_t bytes, const std::string& name) {
    const size_t G = guard_bytes();
    void* p = nullptr;
    if (!G) return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
    const size_t user = (bytes + 255) & ~(size_t)255;             // the zone after the buffer starts at the next 256-byte boundary
    if (hipMalloc(&p, user + 2 * G) != hipSuccess) return nullptr;
    char* base = static_cast<char*>(p);
    if (hipMemset(base, 0xA5, G) != hipSuccess || hipMemset(base + G + bytes, 0xA5, user - bytes + G) != hipSuccess) { (void)hipFree(p); return nullptr; }
    guarded[base + G] = Guarded{base, bytes, name};
    return base + G;
}

void ttc_ctx::guarded_free(void* user) {
    auto it = guarded.find(user);
    if (it == guarded.end()) { (void)hipFree(user); return; }
    (void)hipFree(it->second.base);
    guarded.erase(it);
}

float* ttc_ctx::alloc_f(size_t n, const char* name) {
    void* p = guarded_malloc(n * sizeof(float), name ? name : "alloc#" + std::to_string(allocs.size()));
    if (!p) return nullptr;
    allocs.push_back(p);
    dev_bytes += n * sizeof(float);
    if (name) named[name] = {static_cast<float*>(p), n};
    return static_cast<float*>(p);
}

bool ttc_ctx::alloc_b16(B16& b, size_t units) {
    b.hi = reinterpret_cast<uint4*>(alloc_f(units * 4));
    b.lo = reinterpret_cast<uint4*>(alloc_f(units * 4));
    return b.hi && b.lo;
}

void* ttc_ctx::scratch_buf(const std::string& key, size_t bytes) {
    auto it = scratch.find(key);
    if (it != scratch.end() && it->second.second >= bytes) return it->second.first;
    if (it != scratch.end()) { guarded_free(it->second.first); dev_bytes -= it->second.second; }
    void* p = guarded_malloc(bytes, "scratch:" + key);
    if (!p) { scratch.erase(key); return nullptr; }
    scratch[key] = {p, bytes};
    dev_bytes += bytes;
    return p;
}

void* ttc_ctx::pinned_buf(const std::string& key, size_t bytes) {
    auto it = pinned.find(key);
    if (it != pinned.end() && it->second.second >= bytes) return it->second.first;
    if (it != pinned.end()) (void)hipHostFree(it->second.first);
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { pinned.erase(key); return nullptr; }
    pinned[key] = {p, bytes};
    return p;
}

ttc_status model_alloc(ttc_ctx* c) {
    const Geo g(c->cfg);
    const size_t N = c->cfg.max_windows, N2 = 2 * N;
    const size_t PP = (size_t)g.y.np * g.x.np, P = (size_t)g.y.n * g.x.n;
    const int F = c->cfg.base_filters, Hd = c->cfg.hidden, C = c->cfg.n_bands;
    if (c->cfg.win_in % 4 != 0 || c->cfg.win_in < 28) return c->fail(TTC_ERR_ARG, "win_in must be a multiple of 4 and >= 28");
    if (model_rows(c->cfg) % 4 != 0 || model_rows(c->cfg) < 28) return c->fail(TTC_ERR_ARG, "win_rows must be a multiple of 4 and >= 28 (or 0)");
    if (F != 64 || Hd != 32 || C != 17) return c->fail(TTC_ERR_ARG, "only base_filters=64, hidden=32, n_bands=17 are built");
    auto area = [&](int Axis::*m, int pad) { return (size_t)(g.y.*m + pad) * (g.x.*m + pad); };
    auto rawa = [&](int Axis::*m) { return (size_t)(g.y.*m) * (g.x.*m + 2); };     // raw conv output: rows keep the input pitch
    const size_t Pc1 = area(&Axis::c1, 0), Pc2 = area(&Axis::c2, 0), Pu2 = area(&Axis::u2, 0), Pu2p = area(&Axis::u2, 2);
    const size_t Pu3 = area(&Axis::u3, 0), Pu3p = area(&Axis::u3, 2), Po = area(&Axis::o, 0);
    const size_t Ph2 = (size_t)(g.y.c1 / 2) * (g.x.c1 / 2);
#define A(field, count, name)                                                        \
    if (!(c->field = c->alloc_f((count), name))) return c->fail(TTC_ERR_NOMEM, "hipMalloc " name)
#define B(field, units, name)                                                        \
    if (!c->alloc_b16(c->field, (units))) return c->fail(TTC_ERR_NOMEM, "hipMalloc " name)
    const bool half = c->half();
    A(frames, N * (g.L + 1) * C * PP, "frames");
    const size_t Pr = (size_t)g.y.n * g.x.np;
    A(yg, N2 * 2 * Hd * Pr, "yg"); A(ug, N2 * Hd * P, "u"); A(yc, N2 * Hd * Pr, "yc");
    A(y_med, N * F * Pr, "y_med"); A(y_cat, N * F * Pr, "y_cat");
    A(y_c1, N * 2 * F * rawa(&Axis::c1), "y_c1"); A(y_c2, N * 4 * F * rawa(&Axis::c2), "y_c2");
    A(y_u2, N * 2 * F * rawa(&Axis::u2), "y_u2"); A(y_u2o, N * 2 * F * rawa(&Axis::u2), "y_u2o");
    A(y_u3, N * F * rawa(&Axis::u3), "y_u3"); A(y_out, N * F * rawa(&Axis::o), "y_out");
    if (!half) {
        A(h[0], N2 * Hd * PP, "h0"); A(h[1], N2 * Hd * PP, "h1"); A(rh, N2 * Hd * PP, "rh");
        A(gru_out, N * F * PP, "gru_out");
        A(z_med, N * F * PP, "z_med");
        A(p1, N * F * (P / 4), "p1");
        A(p2, N * 2 * F * Ph2, "p2");
        A(u2in, N * 4 * F * Pu2p, "u2in");
        A(u2a, N * 4 * F * Pu2p, "u2cat");                         // [up2 | crop(conv1)] concat buffer
        A(u3in, N * 2 * F * Pu3p, "u3in");
        A(oa, N * 2 * F * Pu3, "ocat");                            // [up3 | crop(concat)] concat buffer
    } else {
        // channel-blocked hi / lo pairs, 16-byte units = channel blocks x positions
        B(frames16, N * (g.L + 1) * ((C + 7) / 8) * PP, "frames16");
        B(h16[0], N2 * (Hd / 8) * PP, "h16_0"); B(h16[1], N2 * (Hd / 8) * PP, "h16_1"); B(rh16, N2 * (Hd / 8) * PP, "rh16");
        B(gru16, N * (F / 8) * PP, "gru16"); B(z_med16, N * (F / 8) * PP, "z_med16");
        B(p1_16, N * (F / 8) * (P / 4), "p1_16"); B(p2_16, N * (2 * F / 8) * Ph2, "p2_16");
        B(u2in16, N * (4 * F / 8) * Pu2p, "u2in16"); B(u2a16, N * (4 * F / 8) * Pu2p, "u2cat16");
        B(u3in16, N * (2 * F / 8) * Pu3p, "u3in16"); B(oa16, N * (2 * F / 8) * Pu3, "ocat16");
    }
    // GroupNorm partial sums: [n][Cout / 4][slots][2]; the Winograd kernels of the fp32 engine keep one slot per 16 x 8 region
    // and contributing wave (more than the direct kernel's one per 512 positions and wave): size for the larger
    c->stats_floats = N2 * 16 * (size_t)std::max({conv_stat_slots(g.y.np, g.x.np), conv_wino_stat_slots(g.y.np, g.x.np, 64), conv_wino4_stat_slots(g.y.np, g.x.np)}) * 2 + 1024;
    A(stats, c->stats_floats, "stats");
    A(gn, 10 * N2 * 32, "gn");
#undef B
#undef A
    return TTC_OK;
}

// ---------------------------------------------------------------------------------------
static const ttc_tensor* find_t(const ttc_tensor* t, int n, const std::string& name) {
    for (int i = 0; i < n; ++i) if (name == t[i].name) return &t[i];
    return nullptr;
}

static ttc_status upload_conv(ttc_ctx* c, PackedConv& pc, const float* const* hwio, int nsets, int Cin, int Cout, int C0 = -1) {
    if (!model_transposed(c->cfg)) return conv_upload(c, pc, hwio, nsets, Cin, Cout, conv_pick_bn(Cout), C0);
    // transposed planes: tap (kh, kw) of the HWIO kernel acts as (kw, kh)
    const size_t blk = (size_t)Cin * Cout;
    std::vector<std::vector<float>> tw(nsets, std::vector<float>(9 * blk));
    std::vector<const float*> ptr(nsets);
    for (int sidx = 0; sidx < nsets; ++sidx) {
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                std::copy(hwio[sidx] + (size_t)(kw * 3 + kh) * blk, hwio[sidx] + (size_t)(kw * 3 + kh + 1) * blk,
                          tw[sidx].begin() + (size_t)(kh * 3 + kw) * blk);
        ptr[sidx] = tw[sidx].data();
    }
    return conv_upload(c, pc, ptr.data(), nsets, Cin, Cout, conv_pick_bn(Cout), C0);
}

ttc_status model_load(ttc_ctx* c, const ttc_tensor* t, int n) {
    const int Hd = c->cfg.hidden, Cx = c->cfg.n_bands;
    auto need = [&](const std::string& name, int64_t numel) -> const ttc_tensor* {
        const ttc_tensor* p = find_t(t, n, name);
        if (!p) { c->fail(TTC_ERR_ARG, "missing weight tensor: " + name); return nullptr; }
        int64_t e = 1;
        for (int i = 0; i < p->ndim; ++i) e *= p->shape[i];
        if (e != numel) { c->fail(TTC_ERR_ARG, "wrong element count for " + name); return nullptr; }
        return p;
    };
    const char* dirs[2] = {"fw", "bw"};
    const float* gk[2]; const float* ck[2];
    std::vector<float> small;
    c->small_off.clear();
    for (int d = 0; d < 2; ++d) {
        const std::string p = std::string("gru/") + dirs[d] + "/";
        const ttc_tensor* a = need(p + "gates/kernel", 9LL * (Cx + Hd) * 2 * Hd);
        const ttc_tensor* b = need(p + "candidate/kernel", 9LL * (Cx + Hd) * Hd);
        if (!a || !b) return TTC_ERR_ARG;
        gk[d] = a->data; ck[d] = b->data;
        c->small_off[p] = (long)small.size();
        const char* vecs[7] = {"gates_r/gamma", "gates_r/beta", "gates_u/gamma", "gates_u/beta",
                               "candidate/kernel_1", "candidate_y/gamma", "candidate_y/beta"};
        for (const char* v : vecs) {
            const ttc_tensor* q = need(p + v, Hd);
            if (!q) return TTC_ERR_ARG;
            small.insert(small.end(), q->data, q->data + Hd);
        }
    }
    const uint32_t one = c->cfg.one_term_layers;         // 16-bit engine: layers that multiply hi x hi only
    const uint32_t two = c->cfg.precision == 2 ? c->cfg.two_term_layers : 0u;     // fp16 engine: layers that multiply x_hi x (w_hi + w_lo)
    auto terms_of = [&](int bit) { return ((one >> bit) & 1u) ? 1 : (((two >> bit) & 1u) ? 2 : 3); };
    TTC_CHECK(upload_conv(c, c->w_gates, gk, 2, Cx + Hd, 2 * Hd, Cx));
    TTC_CHECK(upload_conv(c, c->w_cand, ck, 2, Cx + Hd, Hd, Cx));
    c->w_gates.terms = terms_of(0);
    c->w_cand.terms = terms_of(1);
    for (int b = 0; b < 8; ++b) {
        const std::string p = std::string(kBlockNames[b]) + "/";
        const int Ci = kBlockCin[b], Co = kBlockCout[b];
        const ttc_tensor* k = need(p + "kernel", 9LL * Ci * Co);
        const ttc_tensor* ga = need(p + "gamma", Co);
        const ttc_tensor* be = need(p + "beta", Co);
        const ttc_tensor* sw = need(p + "sse_kernel", Co);
        const ttc_tensor* sb = need(p + "sse_bias", 1);
        if (!k || !ga || !be || !sw || !sb) return TTC_ERR_ARG;
        const float* kk[1] = {k->data};
        TTC_CHECK(upload_conv(c, c->w_block[b], kk, 1, Ci, Co));
        c->w_block[b].terms = terms_of(2 + b);
        c->small_off[p] = (long)small.size();
        small.insert(small.end(), ga->data, ga->data + Co);
        small.insert(small.end(), be->data, be->data + Co);
        small.insert(small.end(), sw->data, sw->data + Co);
        small.push_back(sb->data[0]);
    }
    const ttc_tensor* hk = need("head/kernel", 64);
    const ttc_tensor* hb = need("head/bias", 1);
    if (!hk || !hb) return TTC_ERR_ARG;
    c->small_off["head/"] = (long)small.size();
    small.insert(small.end(), hk->data, hk->data + 64);
    small.push_back(hb->data[0]);
    if (!c->d_small) {
        c->d_small = c->alloc_f(small.size() + 64);
        if (!c->d_small) return c->fail(TTC_ERR_NOMEM, "hipMalloc small params");
    }
    TTC_HIP(c, hipMemcpy(c->d_small, small.data(), small.size() * sizeof(float), hipMemcpyHostToDevice));
    c->have_model = true;
    return TTC_OK;
}

// ---------------------------------------------------------------------------------------
ttc_status model_frames_from_nhwc(ttc_ctx* c, const float* d_in, int n, hipStream_t s) {
    const Geo g(c->cfg);
    KTimer kt(c, "frames_from_nhwc", s);
    dim3 grid((g.y.np * g.x.np + 255) / 256, g.L + 1, n);
    c->frames_planar_valid = true;
    hipLaunchKernelGGL(k_nhwc_to_frames, grid, dim3(256), 0, s, d_in, c->frames, g.L + 1, g.y.n, g.x.n, c->cfg.n_bands, g.tr ? 1 : 0);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

static ttc_status gn_fin(ttc_ctx* c, float* gn, int nseq, int Cout, int groups, int nblk, double count, hipStream_t s) {
    KTimer kt(c, "gn_finalize", s);
    const int nquads = Cout / 4;
    hipLaunchKernelGGL(k_gn_finalize, dim3(groups, nseq), dim3(64), 0, s, c->stats, gn, nquads, nblk, nquads / groups,
                       count, 1e-5f);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

static ttc_status finalize(ttc_ctx* c, int mode, const FinArgs& a, int n, hipStream_t s) {
    KTimer kt(c, "block_finalize", s);
    dim3 grid((a.Hd * a.Wd + 255) / 256, n);
    const size_t lds = 3 * a.C * sizeof(float);
    if (mode == G_COPY && a.C == 64) hipLaunchKernelGGL((k_block_finalize_1p<G_COPY, 64>), grid, dim3(256), 0, s, a);
    else if (mode == G_COPY && a.C == 128) hipLaunchKernelGGL((k_block_finalize_1p<G_COPY, 128>), grid, dim3(256), 0, s, a);
    else if (mode == G_UP && a.C == 128) hipLaunchKernelGGL((k_block_finalize_1p<G_UP, 128>), grid, dim3(256), 0, s, a);
    else if (mode == G_COPY) hipLaunchKernelGGL(k_block_finalize<G_COPY>, grid, dim3(256), lds, s, a);
    else if (mode == G_POOL) hipLaunchKernelGGL(k_block_finalize<G_POOL>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_block_finalize<G_UP>, grid, dim3(256), lds, s, a);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// ---------------------------------------------------------------------------------------
// forward on the 16-bit engine: same graph, same launch order; every conv input is a channel-blocked hi / lo pair
template <int BF>
static ttc_status forward_h16(ttc_ctx* c, int n, float* d_out, hipStream_t s, FramesForm form) {
    const Geo g(c->cfg);
    const int N = n, N2 = 2 * n, Hd = c->cfg.hidden, Cx = c->cfg.n_bands, F = c->cfg.base_filters;
    const int H = g.y.n, W = g.x.n, Hp = g.y.np, Wp = g.x.np;
    const long PP = (long)Hp * Wp, P = (long)H * W, Pr = (long)H * Wp;
    const int Cx8 = (Cx + 7) / 8, Hd8 = Hd / 8;
    const float* sm = c->d_small;
    float* gn_slot[10];
    for (int i = 0; i < 10; ++i) gn_slot[i] = c->gn + (size_t)i * c->cfg.max_windows * 2 * 32;
    const int nblk_full = conv_stat_slots(Hp, Wp);

    // frames (fp32 planar from the NHWC entry points / the border assembly) -> channel-blocked hi / lo; the tile path's window
    // assembly writes the blocked form itself (tile.hip k_assemble)
    if (form == FRAMES_PLANAR) {
        KTimer kt(c, "frames_to_b16", s);
        hipLaunchKernelGGL((k_planar_to_b16<BF>), dim3((unsigned)((PP + 255) / 256), N * (g.L + 1)), dim3(256), 0, s, c->frames,
                           Cx, PP, Cx8, c->frames16.hi, c->frames16.lo);
        TTC_HIP(c, hipGetLastError());
    }
    // raw (pre-GroupNorm) outputs: exact fp32 as channel-blocked top / bottom half planes in the buffer a planar tensor of the same
    // shape would fill (top plane first, h16_common.h Raw16); `plane` = rows * input pitch
    auto raw_of = [](float* buf, int nseq, int C, long plane) {
        uint4* top = reinterpret_cast<uint4*>(buf);
        return Raw16{top, top + (long)nseq * (C / 8) * plane};
    };
    auto launch = [&](H16Args& a, const PackedConv& pw, int epi, int nseq, float* raw) -> ttc_status {
        if (a.seg[0].C8 + a.seg[1].C8 != pw.nchunk_h) return c->fail(TTC_ERR_STATE, "16-bit conv: channel blocks do not match the packed weights");
        a.nchunk_pack = pw.nchunk_h;
        a.nchunk = a.nchunk ? a.nchunk : pw.nchunk_h;     // a caller may ask for the first segment's chunks only (second segment == 0)
        a.w = pw.d_wh; a.w_set_stride = pw.nsets > 1 ? pw.set_stride_h : 0;
        const long plane = (long)(a.c.Hp - 2) * a.c.Wp;
        const Raw16 r = raw_of(raw, nseq, a.c.Cout, plane);
        a.o_hi = const_cast<uint4*>(r.top); a.o_lo = const_cast<uint4*>(r.bot);
        a.o_stride_n = (long)(a.c.Cout / 8) * plane; a.o_plane = plane;
        TTC_HIP(c, conv_launch_h16(a, pw, BF, epi, OUT_B16, nseq, s));
        return TTC_OK;
    };

    // ---------------- bi-directional ConvGRU ----------------
    // Step 0 of both directions starts from h = 0: the gates and candidate convolutions run over the frame segment's chunks only
    // (3 of 7; channel blocks are per segment, so the state tensors are not touched at all), r * h is never formed and the
    // state buffers are neither cleared nor read -- the same sums minus terms that are exactly zero.
    const GruParams gp{sm + c->small_off["gru/fw/"], c->small_off["gru/bw/"] - c->small_off["gru/fw/"]};
    int cur = 0;
    for (int st = 0; st < g.L; ++st) {
        H16Args a{};
        a.seg[0] = {c->frames16.hi, c->frames16.lo, (long)(g.L + 1) * Cx8 * PP, {(long)st * Cx8 * PP, (long)(g.L - 1 - st) * Cx8 * PP}, Cx8};
        a.seg[1] = {c->h16[cur].hi, c->h16[cur].lo, (long)Hd8 * PP, {0, (long)N * Hd8 * PP}, Hd8};
        a.c.Hp = Hp; a.c.Wp = Wp; a.c.Cout = 2 * Hd; a.c.n_per_set = N;
        a.c.stats = c->stats;
        const bool h0 = st == 0;
        a.nchunk = h0 ? Cx8 : 0;
        { KTimer kt(c, "conv_gates", s); TTC_CHECK(launch(a, c->w_gates, EPI_RAW, N2, c->yg)); kt.flops(conv_issued_flops_h16(a, c->w_gates, N2)); }
        TTC_CHECK(gn_fin(c, gn_slot[8], N2, 2 * Hd, 16, nblk_full, 4.0 * P, s));
        if (!h0) {
            KTimer kt(c, "gru_apply1", s);
            hipLaunchKernelGGL((k_gru_apply1_b16<BF>), dim3((PP + 255) / 256, N2), dim3(256), 0, s, raw_of(c->yg, N2, 2 * Hd, Pr), gn_slot[8], gp,
                               c->h16[cur], c->rh16, H, W, N);
            TTC_HIP(c, hipGetLastError());
        }
        a.seg[1].hi = c->rh16.hi; a.seg[1].lo = c->rh16.lo;
        a.nchunk = h0 ? Cx8 : 0;
        a.c.Cout = Hd;
        a.c.aux = gp.base + 4 * 32; a.c.aux_set_stride = gp.dir_stride;
        { KTimer kt(c, "conv_cand", s); TTC_CHECK(launch(a, c->w_cand, EPI_SSE, N2, c->yc)); kt.flops(conv_issued_flops_h16(a, c->w_cand, N2)); }
        TTC_CHECK(gn_fin(c, gn_slot[9], N2, Hd, 8, nblk_full, 4.0 * P, s));
        {
            KTimer kt(c, "gru_apply2", s);
            hipLaunchKernelGGL((k_gru_apply2_b16<BF>), dim3((PP + 255) / 256, N2), dim3(256), 0, s, raw_of(c->yc, N2, Hd, Pr), gn_slot[9], gp,
                               raw_of(c->yg, N2, 2 * Hd, Pr), gn_slot[8], c->keep_debug ? c->ug : nullptr, c->h16[cur], st == g.L - 1 ? B16{} : c->h16[cur ^ 1],
                               st == g.L - 1 ? c->gru16 : B16{}, H, W, N, c->cfg.zoneout, h0 ? 1 : 0);
            TTC_HIP(c, hipGetLastError());
        }
        cur ^= 1;
    }

    // ---------------- U-Net ----------------
    struct Dim { int h, w; long area() const { return (long)h * w; } };
    auto dim = [&](int Axis::*m, int pad) { return Dim{g.y.*m + pad, g.x.*m + pad}; };
    auto block_conv = [&](int b, H16Seg s0, H16Seg s1, Dim in, int same, float* out, const char* tname) -> ttc_status {
        H16Args a{};
        a.seg[0] = s0; a.seg[1] = s1;
        a.c.Hp = in.h; a.c.Wp = in.w; a.c.Cout = kBlockCout[b]; a.c.n_per_set = N;
        const long Po = (long)(in.h - 2) * (in.w - 2);
        a.c.stats = c->stats; a.c.same_pad = same;
        { KTimer kt(c, tname, s); TTC_CHECK(launch(a, c->w_block[b], EPI_SWISH, N, out)); kt.flops(conv_issued_flops_h16(a, c->w_block[b], N)); }
        return gn_fin(c, gn_slot[b], N, a.c.Cout, 8, conv_stat_slots(in.h, in.w), (double)(a.c.Cout / 8) * Po, s);
    };
    auto prm = [&](int b) { return sm + c->small_off[std::string(kBlockNames[b]) + "/"]; };
    // src: raw conv output dims; dst: blocked destination, dims INCLUDING pad; cblk: its channel blocks per n; boff: block offset
    auto fin = [&](int b, int mode, const float* y, Dim src, const B16& dst, Dim d, int pad, int crop, int cblk, int boff) -> ttc_status {
        FinArgs f{y, gn_slot[b], prm(b), nullptr, kBlockCout[b], src.h, src.w, d.h, d.w, pad, crop, mode, (long)cblk * d.area(), boff,
                  dst.hi, dst.lo, N};
        KTimer kt(c, "block_finalize", s);
        dim3 grid((f.Hd * f.Wd + 255) / 256, N);
        const size_t lds = 3 * f.C * sizeof(float);
        if (mode == G_COPY && f.C == 64) hipLaunchKernelGGL((k_block_finalize_b16_1p<BF, G_COPY, 64>), grid, dim3(256), 0, s, f);
        else if (mode == G_COPY && f.C == 128) hipLaunchKernelGGL((k_block_finalize_b16_1p<BF, G_COPY, 128>), grid, dim3(256), 0, s, f);
        else if (mode == G_UP && f.C == 128) hipLaunchKernelGGL((k_block_finalize_b16_1p<BF, G_UP, 128>), grid, dim3(256), 0, s, f);
        else if (mode == G_COPY) hipLaunchKernelGGL((k_block_finalize_b16<BF, G_COPY>), grid, dim3(256), lds, s, f);
        else if (mode == G_POOL) hipLaunchKernelGGL((k_block_finalize_b16<BF, G_POOL>), grid, dim3(256), lds, s, f);
        else hipLaunchKernelGGL((k_block_finalize_b16<BF, G_UP>), grid, dim3(256), lds, s, f);
        TTC_HIP(c, hipGetLastError());
        return TTC_OK;
    };
    auto seg = [](const B16& t, long stride_n, long off, int C8) { return H16Seg{t.hi, t.lo, stride_n, {off, off}, C8}; };
    const H16Seg none{nullptr, nullptr, 0, {0, 0}, 0};
    const int F8 = F / 8;
    const Dim full{H, W}, fullp{Hp, Wp}, half{H / 2, W / 2};
    const Dim c1 = dim(&Axis::c1, 0), c2 = dim(&Axis::c2, 0), u2 = dim(&Axis::u2, 0), u2p = dim(&Axis::u2, 2);
    const Dim u3 = dim(&Axis::u3, 0), u3p = dim(&Axis::u3, 2), o = dim(&Axis::o, 0);
    const Dim h2{c1.h / 2, c1.w / 2};
    // conv_median on the median frame (zero-padded SAME)
    TTC_CHECK(block_conv(0, seg(c->frames16, (long)(g.L + 1) * Cx8 * PP, (long)g.L * Cx8 * PP, Cx8), none, fullp, 1, c->y_med, "conv_median"));
    TTC_CHECK(fin(0, G_COPY, c->y_med, full, c->z_med16, fullp, 1, 0, F8, 0));
    // conv_concat([gru, median_conv])
    TTC_CHECK(block_conv(1, seg(c->gru16, (long)F8 * PP, 0, F8), seg(c->z_med16, (long)F8 * PP, 0, F8), fullp, 1, c->y_cat, "conv_concat"));
    TTC_CHECK(fin(1, G_POOL, c->y_cat, full, c->p1_16, half, 0, 0, F8, 0));
    // conv1 (VALID) on pool1
    TTC_CHECK(block_conv(2, seg(c->p1_16, (long)F8 * half.area(), 0, F8), none, half, 0, c->y_c1, "conv1"));
    TTC_CHECK(fin(2, G_POOL, c->y_c1, c1, c->p2_16, h2, 0, 0, 2 * F8, 0));
    // conv2 (VALID) on pool2
    TTC_CHECK(block_conv(3, seg(c->p2_16, 2L * F8 * h2.area(), 0, 2 * F8), none, h2, 0, c->y_c2, "conv2"));
    TTC_CHECK(fin(3, G_UP, c->y_c2, c2, c->u2in16, u2p, 1, 0, 4 * F8, 0));
    // up2 (SAME) on nearest x2
    TTC_CHECK(block_conv(4, seg(c->u2in16, 4L * F8 * u2p.area(), 0, 4 * F8), none, u2p, 1, c->y_u2, "conv_up2"));
    TTC_CHECK(fin(4, G_COPY, c->y_u2, u2, c->u2a16, u2p, 1, 0, 4 * F8, 0));
    TTC_CHECK(fin(2, G_COPY, c->y_c1, c1, c->u2a16, u2p, 1, 2, 4 * F8, 2 * F8));            // crop(conv1, 2)
    TTC_CHECK(block_conv(5, seg(c->u2a16, 4L * F8 * u2p.area(), 0, 4 * F8), none, u2p, 1, c->y_u2o, "conv_up2_out"));
    TTC_CHECK(fin(5, G_UP, c->y_u2o, u2, c->u3in16, u3p, 1, 0, 2 * F8, 0));
    // up3 (SAME)
    TTC_CHECK(block_conv(6, seg(c->u3in16, 2L * F8 * u3p.area(), 0, 2 * F8), none, u3p, 1, c->y_u3, "conv_up3"));
    TTC_CHECK(fin(6, G_COPY, c->y_u3, u3, c->oa16, u3, 0, 0, 2 * F8, 0));
    TTC_CHECK(fin(1, G_COPY, c->y_cat, full, c->oa16, u3, 0, 6, 2 * F8, F8));               // crop(concat, 6)
    // out (VALID)
    TTC_CHECK(block_conv(7, seg(c->oa16, 2L * F8 * u3.area(), 0, 2 * F8), none, u3, 0, c->y_out, "out_conv"));
    {
        KTimer kt(c, "head", s);
        const int Po = (int)o.area();
        hipLaunchKernelGGL(k_head<true>, dim3((Po + 255) / 256, N), dim3(256), 4 * F * sizeof(float), s, c->y_out, gn_slot[7],
                           prm(7), sm + c->small_off["head/"], d_out, F, Po, o.w, g.tr ? 1 : 0, N);
        TTC_HIP(c, hipGetLastError());
    }
    c->forward_n = N;
    return TTC_OK;
}

// products per conv layer of the 16-bit engine, switchable after load: the packed weight images hold hi | lo for every layer and every
// producer writes both halves of its output, so a layer's product count is only the kernel instantiation its launch picks (ttc_calibrate_precision)
ttc_status model_set_terms(ttc_ctx* c, uint32_t one, uint32_t two) {
    if (!c->half()) return c->fail(TTC_ERR_ARG, "set_terms: not a 16-bit-engine context");
    if (c->cfg.precision != 2) two = 0;
    c->cfg.one_term_layers = one;
    c->cfg.two_term_layers = two;
    auto terms_of = [&](int bit) { return ((one >> bit) & 1u) ? 1 : (((two >> bit) & 1u) ? 2 : 3); };
    c->w_gates.terms = terms_of(0);
    c->w_cand.terms = terms_of(1);
    for (int b = 0; b < 8; ++b) c->w_block[b].terms = terms_of(2 + b);
    return TTC_OK;
}

ttc_status model_forward_frames(ttc_ctx* c, int n, float* d_out, hipStream_t s, FramesForm form) {
    if (!c->have_model) return c->fail(TTC_ERR_STATE, "ttc_load_weights has not been called");
    if (n <= 0 || n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "window count exceeds max_windows");
    if (c->half()) {
        const int m = c->blk_mode();
        return m == 1 ? forward_h16<1>(c, n, d_out, s, form) : forward_h16<0>(c, n, d_out, s, form);
    }
    if (form != FRAMES_PLANAR) return c->fail(TTC_ERR_STATE, "blocked 16-bit frames handed to the fp32 engine");
    const Geo g(c->cfg);
    const int N = n, N2 = 2 * n, Hd = c->cfg.hidden, Cx = c->cfg.n_bands, F = c->cfg.base_filters;
    const int H = g.y.n, W = g.x.n, Hp = g.y.np, Wp = g.x.np;
    const long PP = (long)Hp * Wp, P = (long)H * W, Pr = (long)H * Wp;
    const float* sm = c->d_small;
    float* gn_slot[10];
    for (int i = 0; i < 10; ++i) gn_slot[i] = c->gn + (size_t)i * c->cfg.max_windows * 2 * 32;

    // ---------------- bi-directional ConvGRU ----------------
    // Step 0 of both directions starts from h = 0: with the Winograd kernels (which can skip channels) the gates and candidate
    // convolutions run over the 17 frame channels only (3 of 7 chunks), r * h is never formed and the state buffer is neither
    // cleared nor read -- the same sums, minus terms that are exactly zero.
    const bool skip_h0 = conv_kernel_for(c->w_gates, EPI_RAW, Hp, Wp, Cx + Hd, N2, N) != CONV_DIRECT &&
                         conv_kernel_for(c->w_cand, EPI_SSE, Hp, Wp, Cx + Hd, N2, N) != CONV_DIRECT;
    if (!skip_h0) TTC_HIP(c, hipMemsetAsync(c->h[0], 0, (size_t)N2 * Hd * PP * sizeof(float), s));
    else {
        // the third chunk holds frame channel 16 and state channels 0 .. 6: those seven planes of h (gates) and r * h (candidate) are
        // the only state the step-0 convolutions read -- clear them (61 MB each instead of 279 MB + a full r * h pass)
        const int zc = 8 * ((Cx + 7) / 8) - Cx;
        KTimer kt(c, "zero_state_planes", s);
        const long cnt = (long)zc * PP;
        hipLaunchKernelGGL(k_zero_planes, dim3((unsigned)((cnt + 255) / 256), N2), dim3(256), 0, s, c->h[0], c->rh, (long)Hd * PP, cnt);
        TTC_HIP(c, hipGetLastError());
    }
    const GruParams gp{sm + c->small_off["gru/fw/"], c->small_off["gru/bw/"] - c->small_off["gru/fw/"]};
    int cur = 0;
    for (int st = 0; st < g.L; ++st) {
        ConvArgs a{};
        a.seg[0] = {c->frames, (long)(g.L + 1) * Cx * PP, {(long)st * Cx * PP, (long)(g.L - 1 - st) * Cx * PP}, Cx};
        a.seg[1] = {c->h[cur], (long)Hd * PP, {0, (long)N * Hd * PP}, Hd};
        a.Cin = Cx + Hd; a.Hp = Hp; a.Wp = Wp; a.Cout = 2 * Hd;
        a.w = c->w_gates.d_w; a.w_set_stride = c->w_gates.set_stride; a.n_per_set = N;
        a.out = c->yg; a.out_stride_n = 2L * Hd * Pr; a.out_plane = Pr; a.out_pitch = Wp; a.oy = a.ox = 0;
        a.stats = c->stats;
        const bool h0 = skip_h0 && st == 0;
        a.cin_run = h0 ? Cx : 0;
        { KTimer kt(c, "conv_gates", s); kt.flops(conv_issued_flops(a, c->w_gates, EPI_RAW, N2)); TTC_HIP(c, conv_launch(a, c->w_gates, EPI_RAW, N2, s)); }
        TTC_CHECK(gn_fin(c, gn_slot[8], N2, 2 * Hd, 16, conv_stat_slots_for(c->w_gates, EPI_RAW, Hp, Wp, a.Cin, N2, N), 4.0 * P, s));
        if (!h0) {
            KTimer kt(c, "gru_apply1", s);
            hipLaunchKernelGGL(k_gru_apply1, dim3((PP + 255) / 256, N2), dim3(256), 0, s, c->yg, gn_slot[8], gp,
                               c->h[cur], c->rh, H, W, N);
            TTC_HIP(c, hipGetLastError());
        }
        a.seg[1].base = c->rh;
        a.Cout = Hd; a.w = c->w_cand.d_w; a.w_set_stride = c->w_cand.set_stride;
        a.out = c->yc; a.out_stride_n = (long)Hd * Pr;
        a.aux = gp.base + 4 * 32; a.aux_set_stride = gp.dir_stride;
        { KTimer kt(c, "conv_cand", s); kt.flops(conv_issued_flops(a, c->w_cand, EPI_SSE, N2)); TTC_HIP(c, conv_launch(a, c->w_cand, EPI_SSE, N2, s)); }
        TTC_CHECK(gn_fin(c, gn_slot[9], N2, Hd, 8, conv_stat_slots_for(c->w_cand, EPI_SSE, Hp, Wp, a.Cin, N2, N), 4.0 * P, s));
        {
            KTimer kt(c, "gru_apply2", s);
            hipLaunchKernelGGL(k_gru_apply2, dim3((PP + 255) / 256, N2), dim3(256), 0, s, c->yc, gn_slot[9], gp,
                               c->yg, gn_slot[8], c->keep_debug ? c->ug : nullptr, c->h[cur], st == g.L - 1 ? nullptr : c->h[cur ^ 1],
                               st == g.L - 1 ? c->gru_out : nullptr, H, W, N, c->cfg.zoneout, h0 ? 1 : 0);
            TTC_HIP(c, hipGetLastError());
        }
        cur ^= 1;
    }

    // ---------------- U-Net ----------------
    struct Dim { int h, w; long area() const { return (long)h * w; } };
    auto dim = [&](int Axis::*m, int pad) { return Dim{g.y.*m + pad, g.x.*m + pad}; };
    auto block_conv = [&](int b, ConvSeg s0, ConvSeg s1, Dim in, int same, float* out, const char* tname) -> ttc_status {
        ConvArgs a{};
        a.seg[0] = s0; a.seg[1] = s1; a.Cin = s0.C + s1.C; a.Hp = in.h; a.Wp = in.w; a.Cout = kBlockCout[b];
        a.w = c->w_block[b].d_w; a.w_set_stride = 0; a.n_per_set = N;
        const long Po = (long)(in.h - 2) * (in.w - 2), Pro = (long)(in.h - 2) * in.w;
        a.out = out; a.out_stride_n = (long)a.Cout * Pro; a.out_plane = Pro; a.out_pitch = in.w; a.oy = a.ox = 0;
        a.stats = c->stats; a.same_pad = same;
        { KTimer kt(c, tname, s); kt.flops(conv_issued_flops(a, c->w_block[b], EPI_SWISH, N)); TTC_HIP(c, conv_launch(a, c->w_block[b], EPI_SWISH, N, s)); }
        return gn_fin(c, gn_slot[b], N, a.Cout, 8, conv_stat_slots_for(c->w_block[b], EPI_SWISH, in.h, in.w, a.Cin, N, N), (double)(a.Cout / 8) * Po, s);
    };
    auto prm = [&](int b) { return sm + c->small_off[std::string(kBlockNames[b]) + "/"]; };
    // src: raw conv output dims; dst: destination dims INCLUDING pad
    auto fin = [&](int b, int mode, const float* y, Dim src, float* dst, Dim d, int pad, int crop, long dst_stride_n,
                   int coff) -> ttc_status {
        FinArgs f{y, gn_slot[b], prm(b), dst, kBlockCout[b], src.h, src.w, d.h, d.w, pad, crop, mode, dst_stride_n, coff};
        return finalize(c, mode, f, N, s);
    };
    const ConvSeg none{nullptr, 0, {0, 0}, 0};
    const Dim full{H, W}, fullp{Hp, Wp}, half{H / 2, W / 2};
    const Dim c1 = dim(&Axis::c1, 0), c2 = dim(&Axis::c2, 0), u2 = dim(&Axis::u2, 0), u2p = dim(&Axis::u2, 2);
    const Dim u3 = dim(&Axis::u3, 0), u3p = dim(&Axis::u3, 2), o = dim(&Axis::o, 0);
    const Dim h2{c1.h / 2, c1.w / 2};
    // conv_median on the median frame (zero-padded SAME)
    TTC_CHECK(block_conv(0, {c->frames + (long)g.L * Cx * PP, (long)(g.L + 1) * Cx * PP, {0, 0}, Cx}, none, fullp, 1,
                         c->y_med, "conv_median"));
    TTC_CHECK(fin(0, G_COPY, c->y_med, full, c->z_med, fullp, 1, 0, (long)F * PP, 0));
    // conv_concat([gru, median_conv])
    TTC_CHECK(block_conv(1, {c->gru_out, (long)F * PP, {0, 0}, F}, {c->z_med, (long)F * PP, {0, 0}, F}, fullp, 1,
                         c->y_cat, "conv_concat"));
    TTC_CHECK(fin(1, G_POOL, c->y_cat, full, c->p1, half, 0, 0, (long)F * half.area(), 0));
    // conv1 (VALID) on pool1
    TTC_CHECK(block_conv(2, {c->p1, (long)F * half.area(), {0, 0}, F}, none, half, 0, c->y_c1, "conv1"));
    TTC_CHECK(fin(2, G_POOL, c->y_c1, c1, c->p2, h2, 0, 0, 2L * F * h2.area(), 0));
    // conv2 (VALID) on pool2
    TTC_CHECK(block_conv(3, {c->p2, 2L * F * h2.area(), {0, 0}, 2 * F}, none, h2, 0, c->y_c2, "conv2"));
    TTC_CHECK(fin(3, G_UP, c->y_c2, c2, c->u2in, u2p, 1, 0, 4L * F * u2p.area(), 0));
    // up2 (SAME) on nearest x2
    TTC_CHECK(block_conv(4, {c->u2in, 4L * F * u2p.area(), {0, 0}, 4 * F}, none, u2p, 1, c->y_u2, "conv_up2"));
    TTC_CHECK(fin(4, G_COPY, c->y_u2, u2, c->u2a, u2p, 1, 0, 4L * F * u2p.area(), 0));
    TTC_CHECK(fin(2, G_COPY, c->y_c1, c1, c->u2a, u2p, 1, 2, 4L * F * u2p.area(), 2 * F));   // crop(conv1, 2)
    TTC_CHECK(block_conv(5, {c->u2a, 4L * F * u2p.area(), {0, 0}, 4 * F}, none, u2p, 1, c->y_u2o, "conv_up2_out"));
    TTC_CHECK(fin(5, G_UP, c->y_u2o, u2, c->u3in, u3p, 1, 0, 2L * F * u3p.area(), 0));
    // up3 (SAME)
    TTC_CHECK(block_conv(6, {c->u3in, 2L * F * u3p.area(), {0, 0}, 2 * F}, none, u3p, 1, c->y_u3, "conv_up3"));
    TTC_CHECK(fin(6, G_COPY, c->y_u3, u3, c->oa, u3, 0, 0, 2L * F * u3.area(), 0));
    TTC_CHECK(fin(1, G_COPY, c->y_cat, full, c->oa, u3, 0, 6, 2L * F * u3.area(), F));     // crop(concat, 6)
    // out (VALID)
    TTC_CHECK(block_conv(7, {c->oa, 2L * F * u3.area(), {0, 0}, 2 * F}, none, u3, 0, c->y_out, "out_conv"));
    {
        KTimer kt(c, "head", s);
        const int Po = (int)o.area();
        hipLaunchKernelGGL(k_head<false>, dim3((Po + 255) / 256, N), dim3(256), 4 * F * sizeof(float), s, c->y_out, gn_slot[7],
                           prm(7), sm + c->small_off["head/"], d_out, F, Po, o.w, g.tr ? 1 : 0, N);
        TTC_HIP(c, hipGetLastError());
        c->forward_n = N;
    }
    return TTC_OK;
}

// the two feature tensors of the forward that has just run for the same n (buffers are still in the workspace)
ttc_status model_taps(ttc_ctx* c, int n, float* d_early, float* d_late, hipStream_t s) {
    if (n <= 0 || n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "window count exceeds max_windows");
    if (n > c->forward_n) return c->fail(TTC_ERR_STATE, "taps of more windows than the last forward ran");
    const Geo g(c->cfg);
    const int F = c->cfg.base_filters;
    KTimer kt(c, "taps", s);
    if (d_early && c->half()) {
        const long total = (long)n * g.y.n * g.x.n * 8;
        const int m = c->blk_mode();
        const dim3 gte((unsigned)((total + 255) / 256));
        if (m == 1) hipLaunchKernelGGL((k_tap_early_b16<1>), gte, dim3(256), 0, s, c->gru16, g.y.n, g.x.n, n, g.tr ? 1 : 0, d_early);
        else hipLaunchKernelGGL((k_tap_early_b16<0>), gte, dim3(256), 0, s, c->gru16, g.y.n, g.x.n, n, g.tr ? 1 : 0, d_early);
    } else if (d_early) {
        const long total = (long)n * g.y.n * g.x.n * 64;
        hipLaunchKernelGGL(k_tap_early, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, c->gru_out, g.y.n, g.x.n, n,
                           g.tr ? 1 : 0, d_early);
    }
    if (d_late) {
        const int Po = g.y.o * g.x.o;
        const float* gn7 = c->gn + (size_t)7 * c->cfg.max_windows * 2 * 32;
        if (c->half()) hipLaunchKernelGGL(k_tap_late<true>, dim3((Po + 255) / 256, n), dim3(256), 3 * F * sizeof(float), s, c->y_out, gn7,
                                          c->d_small + c->small_off["out/"], d_late, F, Po, g.x.o, g.tr ? 1 : 0, c->forward_n);
        else hipLaunchKernelGGL(k_tap_late<false>, dim3((Po + 255) / 256, n), dim3(256), 3 * F * sizeof(float), s, c->y_out, gn7,
                                c->d_small + c->small_off["out/"], d_late, F, Po, g.x.o, g.tr ? 1 : 0, c->forward_n);
    }
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
