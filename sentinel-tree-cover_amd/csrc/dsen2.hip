// 20 m -> 10 m: bilinear pre-upsample (process_tile, src/download_and_predict_job.py:734-782) and the
// DSen2-lite residual CNN (models-release/supres-40k-swir/superresolve_graph.pb) with
// its whole-tile window driver (superresolve_large_tile, job.py:95-147).
//
// The six 3x3 convolutions run on the same fp32-MFMA implicit-GEMM engine as the ConvGRU/U-Net
// (conv3x3_mfma.hip): activations live planar and reflect-padded ([n][32][H+2][W+2]); each conv
// writes the interior of the next padded buffer and k_reflect_border fills the 1-px rim
// (MirrorPad REFLECT before every Conv2D VALID in the graph).
#include "h16_common.h"

namespace {

using B16 = ttc_ctx::B16;

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// NHWC [n][H][W][C] (channels [c0, c0+Cs) of a Cin-wide pixel) -> planar [n][Cs][H+2p][W+2p], reflect pad p
__global__ void k_nhwc_to_planar(const float* __restrict__ in, int Cin, int c0, int Cs, int H, int W, int pad,
                                 float* __restrict__ out) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Hp * Wp) return;
    const int y = reflect_idx(p / Wp - pad, H), x = reflect_idx(p % Wp - pad, W);
    const float* src = in + (((long)n * H + y) * W + x) * Cin + c0;
    float* dst = out + (long)n * Cs * Hp * Wp + p;
    for (int c = 0; c < Cs; ++c) dst[(long)c * Hp * Wp] = src[c];
}

// fill the 1-px reflect rim of [n*C][Hp][Wp] planes from their interior
__global__ void k_reflect_border(float* __restrict__ buf, int Hp, int Wp) {
    float* pl = buf + (long)blockIdx.y * Hp * Wp;
    const int per = 2 * Wp + 2 * Hp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    int y, x;
    if (i < Wp) { y = 0; x = i; }
    else if (i < 2 * Wp) { y = Hp - 1; x = i - Wp; }
    else if (i < 2 * Wp + Hp) { y = i - 2 * Wp; x = 0; }
    else { y = i - 2 * Wp - Hp; x = Wp - 1; }
    const int sy = 1 + reflect_idx(y - 1, Hp - 2), sx = 1 + reflect_idx(x - 1, Wp - 2);
    pl[y * Wp + x] = pl[sy * Wp + sx];
}

// planar [n][C][H][W] -> NHWC [n][H][W][C]
__global__ void k_planar_to_nhwc(const float* __restrict__ in, int C, int H, int W, float* __restrict__ out) {
    const int n = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    for (int c = 0; c < C; ++c) out[((long)n * H * W + p) * C + c] = in[((long)n * C + c) * H * W + p];
}

struct SRWin { int n; int x0[40], y0[40]; };

// tile [T][X][Y][10] -> padded planar window batch [(t*nw + w)][10][ws+10][ws+10]: window reflect-padded by 4
// (job.py:112) then by the first conv's 1 (MirrorPad)
__global__ void k_sr_gather(const float* __restrict__ tile, int X, int Y, SRWin sw, int ws, int cs, float* __restrict__ out) {
    const int E = ws + 8, Ep = E + 2;
    const int w = blockIdx.y, t = blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Ep * Ep) return;
    const int qy = reflect_idx(p / Ep - 1, E), qx = reflect_idx(p % Ep - 1, E);
    const int lx = reflect_idx(qy - 4, ws), ly = reflect_idx(qx - 4, ws);
    const float* src = tile + (((long)t * X + sw.x0[w] + lx) * Y + sw.y0[w] + ly) * cs;      // cs: floats per pixel (>= 10)
    float* dst = out + ((long)(t * sw.n + w) * 10) * Ep * Ep + p;
    for (int c = 0; c < 10; ++c) dst[(long)c * Ep * Ep] = src[c];
}

// 16-bit engine: the same gather, written channel-blocked (10 bands -> 2 blocks, 6 zero pad channels) as hi / lo pairs
template <int BF>
__global__ void k_sr_gather_b16(const float* __restrict__ tile, int X, int Y, SRWin sw, int ws, int cs, uint4* __restrict__ ohi,
                                uint4* __restrict__ olo) {
    const int E = ws + 8, Ep = E + 2;
    const int w = blockIdx.y, t = blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Ep * Ep) return;
    const int qy = reflect_idx(p / Ep - 1, E), qx = reflect_idx(p % Ep - 1, E);
    const int lx = reflect_idx(qy - 4, ws), ly = reflect_idx(qx - 4, ws);
    const float* src = tile + (((long)t * X + sw.x0[w] + lx) * Y + sw.y0[w] + ly) * cs;
    const long u = ((long)(t * sw.n + w) * 2) * Ep * Ep + p;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = src[c];
    b16_store8<BF>(ohi, olo, u, v);
    v[0] = src[8]; v[1] = src[9];
#pragma unroll
    for (int c = 2; c < 8; ++c) v[c] = 0.0f;
    b16_store8<BF>(ohi, olo, u + (long)Ep * Ep, v);
}
// ... and the exact fp32 bilinear operand straight from the tile (bands 4..9 of the 4-padded window, job.py:114)
__global__ void k_sr_bil_tile(const float* __restrict__ tile, int X, int Y, SRWin sw, int ws, int cs, float* __restrict__ bil) {
    const int E = ws + 8;
    const int w = blockIdx.y, t = blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E * E) return;
    const int lx = reflect_idx(p / E - 4, ws), ly = reflect_idx(p % E - 4, ws);
    const float* src = tile + (((long)t * X + sw.x0[w] + lx) * Y + sw.y0[w] + ly) * cs + 4;
    float* dst = bil + ((long)(t * sw.n + w) * 6) * E * E + p;
    for (int c = 0; c < 6; ++c) dst[(long)c * E * E] = src[c];
}
// fill the 1-px reflect rim of blocked planes [img][Hp*Wp] (degenerate window sizes only)
__global__ void k_reflect_border_b16(uint4* __restrict__ hi, uint4* __restrict__ lo, int Hp, int Wp) {
    const long base = (long)blockIdx.y * Hp * Wp;
    const int per = 2 * Wp + 2 * Hp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    int y, x;
    if (i < Wp) { y = 0; x = i; }
    else if (i < 2 * Wp) { y = Hp - 1; x = i - Wp; }
    else if (i < 2 * Wp + Hp) { y = i - 2 * Wp; x = 0; }
    else { y = i - 2 * Wp - Hp; x = Wp - 1; }
    const int sy = 1 + reflect_idx(y - 1, Hp - 2), sx = 1 + reflect_idx(x - 1, Wp - 2);
    hi[base + y * Wp + x] = hi[base + sy * Wp + sx];
    lo[base + y * Wp + x] = lo[base + sy * Wp + sx];
}

// bilinear operand of the graph = channels 4..9 of the 4-padded window (job.py:114): the interior of
// planes 4..9 of the gathered input
__global__ void k_sr_bil(const float* __restrict__ xin, int E, float* __restrict__ bil) {
    const int Ep = E + 2, n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E * E) return;
    const int y = p / E, x = p % E;
    for (int c = 0; c < 6; ++c)
        bil[((long)n * 6 + c) * E * E + p] = xin[((long)n * 10 + 4 + c) * Ep * Ep + (long)(y + 1) * Ep + x + 1];
}

// planar result [(t*nw + w)][6][E][E] -> centre crop [4:-4] into channels 4..9 of the tile (job.py:118-119)
__global__ void k_sr_scatter(const float* __restrict__ res, int X, int Y, SRWin sw, int ws, int cs, int w_lo, int w_hi,
                             float* __restrict__ tile) {
    const int E = ws + 8;
    const int w = w_lo + blockIdx.y, t = blockIdx.z;
    if (w >= w_hi) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ws * ws) return;
    const int lx = p / ws, ly = p % ws;
    const float* src = res + ((long)(t * sw.n + w) * 6) * E * E + (long)(lx + 4) * E + (ly + 4);
    float* dst = tile + (((long)t * X + sw.x0[w] + lx) * Y + sw.y0[w] + ly) * cs + 4;
    for (int c = 0; c < 6; ++c) dst[c] = src[(long)c * E * E];
}

// ---- bilinear resize as skimage.transform.resize(order=1) evaluates it for upsampling:
// scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True) in float64 (job.py:741-743, :759-781)
__device__ __forceinline__ double bil_sample(const float* __restrict__ src, long stride_y, long stride_x, int h, int w,
                                             double sy, double sx) {
    const int y0 = (int)floor(sy), x0 = (int)floor(sx);
    const double fy = sy - y0, fx = sx - x0;
    const int ya = reflect_idx(y0, h), yb = reflect_idx(y0 + 1, h), xa = reflect_idx(x0, w), xb = reflect_idx(x0 + 1, w);
    const double v00 = src[ya * stride_y + xa * stride_x], v01 = src[ya * stride_y + xb * stride_x];
    const double v10 = src[yb * stride_y + xa * stride_x], v11 = src[yb * stride_y + xb * stride_x];
    // separable evaluation, rows (axis 0) first like ndimage's sequential 1-D passes
    const double a = v00 * (1.0 - fy) + v10 * fy, b = v01 * (1.0 - fy) + v11 * fy;
    return a * (1.0 - fx) + b * fx;
}

__global__ void k_upsample_20m(const float* __restrict__ s10, const float* __restrict__ s20, int h, int w,
                               float* __restrict__ out) {
    const int t = blockIdx.y, H = 2 * h, W = 2 * w;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    float* dst = out + ((long)t * H * W + p) * 10;
    const float* a = s10 + ((long)t * H * W + p) * 4;
    dst[0] = a[0]; dst[1] = a[1]; dst[2] = a[2]; dst[3] = a[3];
    const float* b = s20 + (long)t * h * w * 6;
    const double sy = (y + 0.5) * 0.5 - 0.5, sx = (x + 0.5) * 0.5 - 0.5;
    for (int c = 0; c < 4; ++c) dst[4 + c] = (float)bil_sample(b + c, (long)w * 6, 6, h, w, sy, sx);
}

// 40 m bands (indices 4, 5 of the 20 m stack): 2x2 mean (float32) then bilinear (job.py:754-782).  On odd
// 20 m grids (309 for a 618 tile) the reference sets the first row / column aside (oy / ox = 1), averages
// the rest, resizes it to (2h - oy, 2w - ox) and writes the set-aside row / column back nearest-replicated.
__global__ void k_mean2x2(const float* __restrict__ s20, int h, int w, int oy, int ox, float* __restrict__ m) {
#pragma clang fp contract(off)
    const int t = blockIdx.y, hh = (h - oy) / 2, ww = (w - ox) / 2;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hh * ww) return;
    const int y = p / ww, x = p % ww;
    const float* b = s20 + ((long)t * h * w + (long)(2 * y + oy) * w + 2 * x + ox) * 6;
    for (int c = 0; c < 2; ++c) {
        const float v00 = b[4 + c], v01 = b[6 + 4 + c], v10 = b[(long)w * 6 + 4 + c], v11 = b[(long)w * 6 + 6 + 4 + c];
        // np.mean over axes (1, 3) of the [h/2, 2, w/2, 2] view: float32 accumulate, divide by 4
        m[((long)t * 2 + c) * hh * ww + p] = (((v00 + v01) + v10) + v11) / 4.0f;
    }
}

__global__ void k_upsample_40m(const float* __restrict__ m, const float* __restrict__ s20, int h, int w, int oy, int ox,
                               float* __restrict__ out) {
    const int t = blockIdx.y, H = 2 * h, W = 2 * w, hh = (h - oy) / 2, ww = (w - ox) / 2;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    float* dst = out + ((long)t * H * W + p) * 10;
    const float* b = s20 + (long)t * h * w * 6;
    for (int c = 0; c < 2; ++c) {
        float v;
        if (ox && x == 0) v = b[((long)(y / 2) * w) * 6 + 4 + c];            // first column, written last (job.py:769)
        else if (oy && y == 0) v = b[(long)(x / 2) * 6 + 4 + c];             // first row
        else {
            const double sy = (y - oy + 0.5) * ((double)hh / (H - oy)) - 0.5, sx = (x - ox + 0.5) * ((double)ww / (W - ox)) - 0.5;
            v = (float)bil_sample(m + ((long)t * 2 + c) * hh * ww, ww, 1, hh, ww, sy, sx);
        }
        dst[8 + c] = v;
    }
}

// The single-call tile path's form of the three kernels above, from the bands AS STORED (uint16, tof_downloading.py:51-61): decode
// (x / 65535.0f, to_float32 :64-72), the 10 m copy, the four 20 m bilinear bands and the two 40 m bands (2x2 float32 mean evaluated
// inline in k_mean2x2's order, then the same bilinear) in ONE pass that writes each pixel's 40-byte record once -- no float copies of
// the raw bands (137 MB written + read back per T = 12 tile), no mean plane, one launch instead of five.  Same arithmetic, same order:
// bit-identical to decode -> upsample_20m.
__device__ __forceinline__ float dec16(unsigned short v) { return (float)v / 65535.0f; }
__device__ __forceinline__ float mean4_u16(const unsigned short* __restrict__ b, long w6) {
#pragma clang fp contract(off)
    const float v00 = dec16(b[0]), v01 = dec16(b[6]), v10 = dec16(b[w6]), v11 = dec16(b[w6 + 6]);
    return (((v00 + v01) + v10) + v11) / 4.0f;
}
// am: the 10 m array as stored may be a pixel or two off the 20 m grid; adjust_shape (job.py:260-310, applied at :720) is its index map
__global__ void k_decode_upsample(const unsigned short* __restrict__ s10, const unsigned short* __restrict__ s20, int h, int w, int oy, int ox,
                                  AdjustMap am, float* __restrict__ out) {
    const int t = blockIdx.y, H = 2 * h, W = 2 * w, hh = (h - oy) / 2, ww = (w - ox) / 2;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    float v[10];
    const int ys = min(max(y + am.o1, 0), am.n1 - 1), xs = min(max(x + am.o2, 0), am.n2 - 1);
    const ushort4 a = *reinterpret_cast<const ushort4*>(s10 + (((long)t * am.n1 + ys) * am.n2 + xs) * 4);
    v[0] = dec16(a.x); v[1] = dec16(a.y); v[2] = dec16(a.z); v[3] = dec16(a.w);
    const unsigned short* b = s20 + (long)t * h * w * 6;
    {
        const double sy = (y + 0.5) * 0.5 - 0.5, sx = (x + 0.5) * 0.5 - 0.5;
        const int y0 = (int)floor(sy), x0 = (int)floor(sx);
        const double fy = sy - y0, fx = sx - x0;
        const int ya = reflect_idx(y0, h), yb = reflect_idx(y0 + 1, h), xa = reflect_idx(x0, w), xb = reflect_idx(x0 + 1, w);
        for (int c = 0; c < 4; ++c) {
            const double v00 = dec16(b[((long)ya * w + xa) * 6 + c]), v01 = dec16(b[((long)ya * w + xb) * 6 + c]);
            const double v10 = dec16(b[((long)yb * w + xa) * 6 + c]), v11 = dec16(b[((long)yb * w + xb) * 6 + c]);
            const double r0 = v00 * (1.0 - fy) + v10 * fy, r1 = v01 * (1.0 - fy) + v11 * fy;
            v[4 + c] = (float)(r0 * (1.0 - fx) + r1 * fx);
        }
    }
    for (int c = 0; c < 2; ++c) {
        float r;
        if (ox && x == 0) r = dec16(b[((long)(y / 2) * w) * 6 + 4 + c]);          // first column, written last (job.py:769)
        else if (oy && y == 0) r = dec16(b[(long)(x / 2) * 6 + 4 + c]);           // first row
        else {
            const double sy = (y - oy + 0.5) * ((double)hh / (H - oy)) - 0.5, sx = (x - ox + 0.5) * ((double)ww / (W - ox)) - 0.5;
            const int y0 = (int)floor(sy), x0 = (int)floor(sx);
            const double fy = sy - y0, fx = sx - x0;
            const int ya = reflect_idx(y0, hh), yb = reflect_idx(y0 + 1, hh), xa = reflect_idx(x0, ww), xb = reflect_idx(x0 + 1, ww);
            auto m = [&](int my, int mx) { return (double)mean4_u16(b + ((long)(2 * my + oy) * w + 2 * mx + ox) * 6 + 4 + c, (long)w * 6); };
            const double r0 = m(ya, xa) * (1.0 - fy) + m(yb, xa) * fy, r1 = m(ya, xb) * (1.0 - fy) + m(yb, xb) * fy;
            r = (float)(r0 * (1.0 - fx) + r1 * fx);
        }
        v[8 + c] = r;
    }
    float2* dst = reinterpret_cast<float2*>(out + ((long)t * H * W + p) * 10);
#pragma unroll
    for (int c = 0; c < 5; ++c) dst[c] = make_float2(v[2 * c], v[2 * c + 1]);
}

const char* kDsNames[6] = {"in_conv", "01_conv", "02_conv", "11_conv", "12_conv", "out_conv"};
const int kDsCin[6] = {10, 32, 32, 32, 32, 32};
const int kDsCout[6] = {32, 32, 32, 32, 32, 6};

}  // namespace

ttc_status dsen2_load(ttc_ctx* c, const ttc_tensor* t, int n) {
    std::vector<float> bias;
    for (int l = 0; l < 6; ++l) {
        const ttc_tensor *k = nullptr, *b = nullptr;
        for (int i = 0; i < n; ++i) {
            if (std::string(kDsNames[l]) + "/kernel" == t[i].name) k = &t[i];
            if (std::string(kDsNames[l]) + "/bias" == t[i].name) b = &t[i];
        }
        if (!k || !b) return c->fail(TTC_ERR_ARG, std::string("missing DSen2 tensor for ") + kDsNames[l]);
        PackedConv& pc = c->w_ds[l];
        const float* kk[1] = {k->data};
        // ttc_config.dsen2_precision: an fp32 context may run these six convs on the 16-bit engine (hi + lo pairs, always three products there)
        TTC_CHECK(conv_upload(c, pc, kk, 1, kDsCin[l], kDsCout[l], 32, -1, c->ds_half() && !c->half() ? c->cfg.dsen2_precision : -1));
        pc.terms = (c->half() && ((c->cfg.one_term_layers >> (10 + l)) & 1u)) ? 1 : 3;
        std::vector<float> bb(32, 0.0f);
        for (int i = 0; i < pc.Cout; ++i) bb[i] = b->data[i];
        bias.insert(bias.end(), bb.begin(), bb.end());
    }
    if (!c->d_ds_bias && !(c->d_ds_bias = c->alloc_f(bias.size()))) return c->fail(TTC_ERR_NOMEM, "hipMalloc DSen2 bias");
    TTC_HIP(c, hipMemcpy(c->d_ds_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    c->have_dsen2 = true;
    return TTC_OK;
}

// runs the six convs on a padded planar batch xin [n][10][H+2][W+2]; bil planar [n][6][H][W];
// result planar [n][6][H][W] (res buffer == out layout)
static ttc_status dsen2_core(ttc_ctx* c, const float* xin, const float* bil, int n, int H, int W, float* out,
                             hipStream_t s) {
    const int Hp = H + 2, Wp = W + 2;
    const long PP = (long)Hp * Wp, P = (long)H * W;
    const size_t bytes = sizeof(float) * (size_t)n * 32 * PP;
    float* A = static_cast<float*>(c->scratch_buf("ds_A", bytes));
    float* B = static_cast<float*>(c->scratch_buf("ds_B", bytes));
    float* Cb = static_cast<float*>(c->scratch_buf("ds_C", bytes));
    if (!A || !B || !Cb) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
    auto conv = [&](int l, const float* in, int Cin, int epi, float* dst, const float* res, bool padded_out) -> ttc_status {
        ConvArgs a{};
        a.seg[0] = {in, (long)Cin * PP, {0, 0}, Cin};
        a.seg[1] = {nullptr, 0, {0, 0}, 0};
        a.Cin = Cin; a.Hp = Hp; a.Wp = Wp; a.Cout = kDsCout[l];
        a.w = c->w_ds[l].d_w; a.w_set_stride = 0; a.n_per_set = n;
        a.out = dst; a.res = res; a.aux = c->d_ds_bias + 32 * l;
        // (rim by a separate k_reflect_border pass instead of the epilogue's mirror stores: convs 5.02 -> 4.72 ms per tile, the ten
        // extra launches 0.6 ms -- kept in the epilogue)
        if (padded_out) { a.out_stride_n = 32 * PP; a.out_plane = PP; a.out_pitch = Wp; a.oy = a.ox = 1; a.reflect_out = (H >= 4 && W >= 4); }
        else { a.out_stride_n = 6 * P; a.out_plane = P; a.out_pitch = W; a.oy = a.ox = 0; }
        { KTimer kt(c, "dsen2_conv", s); kt.flops(conv_issued_flops(a, c->w_ds[l], epi, n)); TTC_HIP(c, conv_launch(a, c->w_ds[l], epi, n, s)); }
        if (padded_out && !a.reflect_out) {                  // degenerate sizes only: the epilogue writes the rim otherwise
            KTimer kt(c, "dsen2_border", s);
            hipLaunchKernelGGL(k_reflect_border, dim3((2 * Wp + 2 * Hp + 255) / 256, n * 32), dim3(256), 0, s, dst, Hp, Wp);
            TTC_HIP(c, hipGetLastError());
        }
        return TTC_OK;
    };
    TTC_CHECK(conv(0, xin, 10, EPI_BIAS_RELU, A, nullptr, true));        // x0 = relu(in_conv)
    TTC_CHECK(conv(1, A, 32, EPI_BIAS_RELU, B, nullptr, true));
    TTC_CHECK(conv(2, B, 32, EPI_BIAS_RES, Cb, A, true));                // x1 = x0 + 0.1 * conv
    TTC_CHECK(conv(3, Cb, 32, EPI_BIAS_RELU, B, nullptr, true));
    TTC_CHECK(conv(4, B, 32, EPI_BIAS_RES, A, Cb, true));                // x2 = x1 + 0.1 * conv
    TTC_CHECK(conv(5, A, 32, EPI_BIAS_TANH_ADD, out, bil, false));       // bilinear + tanh(out_conv)
    return TTC_OK;
}

// the six convs on the 16-bit engine: xin blocked [n][2][PP] hi / lo (reflect-padded), intermediates blocked [n][4][PP];
// conv epilogues write the next conv's padded input directly (bias / ReLU / 0.1 x residual fused, reflect rim included)
template <int BF>
static ttc_status dsen2_core_h16(ttc_ctx* c, const B16& xin, const float* bil, int n, int H, int W, float* out, hipStream_t s) {
    const int Hp = H + 2, Wp = W + 2;
    const long PP = (long)Hp * Wp, P = (long)H * W;
    const size_t bytes = (size_t)n * 4 * PP * 16;
    B16 A{static_cast<uint4*>(c->scratch_buf("ds16_Ah", bytes)), static_cast<uint4*>(c->scratch_buf("ds16_Al", bytes))};
    B16 B{static_cast<uint4*>(c->scratch_buf("ds16_Bh", bytes)), static_cast<uint4*>(c->scratch_buf("ds16_Bl", bytes))};
    B16 Cb{static_cast<uint4*>(c->scratch_buf("ds16_Ch", bytes)), static_cast<uint4*>(c->scratch_buf("ds16_Cl", bytes))};
    if (!A.hi || !A.lo || !B.hi || !B.lo || !Cb.hi || !Cb.lo) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
    const bool rim = (H >= 4 && W >= 4);
    auto conv = [&](int l, const B16& in, int C8, int epi, const B16* dst, const B16* res) -> ttc_status {
        const PackedConv& pw = c->w_ds[l];
        H16Args a{};
        a.seg[0] = {in.hi, in.lo, (long)C8 * PP, {0, 0}, C8};
        a.seg[1] = {nullptr, nullptr, 0, {0, 0}, 0};
        a.nchunk = C8;
        if (a.nchunk != pw.nchunk_h) return c->fail(TTC_ERR_STATE, "DSen2 16-bit conv: channel blocks do not match the packed weights");
        a.w = pw.d_wh; a.w_set_stride = 0;
        a.c.Hp = Hp; a.c.Wp = Wp; a.c.Cout = kDsCout[l]; a.c.n_per_set = n;
        a.c.aux = c->d_ds_bias + 32 * l;
        int kind = OUT_F32;
        if (dst) {
            kind = OUT_B16;
            a.o_hi = dst->hi; a.o_lo = dst->lo; a.o_stride_n = 4 * PP; a.o_plane = PP;
            a.c.out_pitch = Wp; a.c.oy = a.c.ox = 1; a.c.reflect_out = rim ? 1 : 0;
            if (res) { a.r_hi = res->hi; a.r_lo = res->lo; }
        } else {
            a.c.out = out; a.c.res = bil; a.c.out_stride_n = 6 * P; a.c.out_plane = P; a.c.out_pitch = W;
        }
        { KTimer kt(c, "dsen2_conv", s); kt.flops(conv_issued_flops_h16(a, pw, n)); TTC_HIP(c, conv_launch_h16(a, pw, BF, epi, kind, n, s)); }
        if (dst && !rim) {
            hipLaunchKernelGGL(k_reflect_border_b16, dim3((2 * Wp + 2 * Hp + 255) / 256, n * 4), dim3(256), 0, s, dst->hi, dst->lo, Hp, Wp);
            TTC_HIP(c, hipGetLastError());
        }
        return TTC_OK;
    };
    TTC_CHECK(conv(0, xin, 2, EPI_BIAS_RELU, &A, nullptr));              // x0 = relu(in_conv)
    TTC_CHECK(conv(1, A, 4, EPI_BIAS_RELU, &B, nullptr));
    TTC_CHECK(conv(2, B, 4, EPI_BIAS_RES, &Cb, &A));                     // x1 = x0 + 0.1 * conv
    TTC_CHECK(conv(3, Cb, 4, EPI_BIAS_RELU, &B, nullptr));
    TTC_CHECK(conv(4, B, 4, EPI_BIAS_RES, &A, &Cb));                     // x2 = x1 + 0.1 * conv
    TTC_CHECK(conv(5, A, 4, EPI_BIAS_TANH_ADD, nullptr, nullptr));       // bilinear + tanh(out_conv), fp32
    return TTC_OK;
}

ttc_status dsen2_forward(ttc_ctx* c, const float* d_in, const float* d_bil, int n, int H, int W, float* d_out,
                         hipStream_t s) {
    if (!c->have_dsen2) return c->fail(TTC_ERR_STATE, "ttc_load_dsen2_weights has not been called");
    if (!d_in || !d_bil || !d_out || n < 1 || H < 3 || W < 3) return c->fail(TTC_ERR_ARG, "dsen2_forward: bad argument");
    const int Hp = H + 2, Wp = W + 2;
    float* xin = static_cast<float*>(c->scratch_buf("ds_in", sizeof(float) * (size_t)n * 10 * Hp * Wp));
    float* bil = static_cast<float*>(c->scratch_buf("ds_bil", sizeof(float) * (size_t)n * 6 * H * W));
    float* res = static_cast<float*>(c->scratch_buf("ds_out", sizeof(float) * (size_t)n * 6 * H * W));
    if (!xin || !bil || !res) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
    hipLaunchKernelGGL(k_nhwc_to_planar, dim3((Hp * Wp + 255) / 256, n), dim3(256), 0, s, d_in, 10, 0, 10, H, W, 1, xin);
    hipLaunchKernelGGL(k_nhwc_to_planar, dim3((H * W + 255) / 256, n), dim3(256), 0, s, d_bil, 6, 0, 6, H, W, 0, bil);
    TTC_HIP(c, hipGetLastError());
    if (c->ds_half()) {
        const size_t ub = (size_t)n * 2 * Hp * Wp * 16;
        B16 x16{static_cast<uint4*>(c->scratch_buf("ds16_inh", ub)), static_cast<uint4*>(c->scratch_buf("ds16_inl", ub))};
        if (!x16.hi || !x16.lo) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
        const dim3 g16((Hp * Wp + 255) / 256, n);
        const int m = c->ds_blk_mode();
        if (m == 1) hipLaunchKernelGGL((k_planar_to_b16<1>), g16, dim3(256), 0, s, xin, 10, (long)Hp * Wp, 2, x16.hi, x16.lo);
        else hipLaunchKernelGGL((k_planar_to_b16<0>), g16, dim3(256), 0, s, xin, 10, (long)Hp * Wp, 2, x16.hi, x16.lo);
        TTC_HIP(c, hipGetLastError());
        TTC_CHECK(m == 1 ? dsen2_core_h16<1>(c, x16, bil, n, H, W, res, s) : dsen2_core_h16<0>(c, x16, bil, n, H, W, res, s));
    } else {
        TTC_CHECK(dsen2_core(c, xin, bil, n, H, W, res, s));
    }
    hipLaunchKernelGGL(k_planar_to_nhwc, dim3((H * W + 255) / 256, n), dim3(256), 0, s, res, 6, H, W, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

// ws: window edge (110 in job.py:121, 125 in resegment_tiles_wide.py:157); cs: floats per pixel of d_s2 (the border strip
// keeps its 4 smoothed indices behind the 10 bands)
ttc_status dsen2_tile(ttc_ctx* c, float* d_s2, int T, int X, int Y, int quirks, int ws, int cs, hipStream_t s) {
    if (!c->have_dsen2) return c->fail(TTC_ERR_STATE, "ttc_load_dsen2_weights has not been called");
    if (ws < 8 || ws > 256 || cs < 10) return c->fail(TTC_ERR_ARG, "superresolve_tile: window edge must be in [8, 256], >= 10 floats per pixel");
    if (!d_s2 || T < 1 || X < ws || Y < ws) return c->fail(TTC_ERR_ARG, "superresolve_tile: tile smaller than one window");
    std::vector<int> xr, yr;
    for (int v = 0; v < X - ws; v += ws) xr.push_back(v);
    xr.push_back(X - ws);
    for (int v = 0; v < Y - ws; v += ws) yr.push_back(v);
    yr.push_back(Y - ws);
    // Reference loop (job.py:131-143): rows x != last read the ORIGINAL tile (their windows are disjoint);
    // the last row reads a private copy x_end that its own calls mutate in place, so its final window
    // (y == last) sees the already-refined columns [Y-110, y_prev+110); windows with y == last and
    // x != last are never run (unreachable elif).  quirks == 0 runs every window on original data.
    // pass-1 windows in four write groups (windows inside a group are disjoint; later groups overwrite
    // earlier ones exactly like the reference's loop order): interior, last column, last row, corner.
    SRWin p1{}, p2{};
    int grp[5] = {0, 0, 0, 0, 0};
    const size_t nx = xr.size(), ny = yr.size();
    for (int g = 0; g < 4; ++g) {
        for (size_t ix = 0; ix < nx; ++ix)
            for (size_t iy = 0; iy < ny; ++iy) {
                const bool lx = ix + 1 == nx, ly = iy + 1 == ny;
                if (g != (lx ? 2 : 0) + (ly ? 1 : 0)) continue;
                if (quirks && ly && !lx) continue;                       // unreachable elif, job.py:141-143
                if (quirks && ly && lx) { p2.x0[0] = xr[ix]; p2.y0[0] = yr[iy]; p2.n = 1; continue; }
                if (p1.n >= 40) return c->fail(TTC_ERR_ARG, "superresolve_tile: more than 40 windows");
                p1.x0[p1.n] = xr[ix]; p1.y0[p1.n] = yr[iy]; p1.n++;
            }
        grp[g + 1] = p1.n;
    }
    if (p1.n > 40) return c->fail(TTC_ERR_ARG, "superresolve_tile: more than 40 windows");
    const int E = ws + 8, Ep = E + 2;
    for (int pass = 0; pass < 2; ++pass) {
        const SRWin& sw = pass == 0 ? p1 : p2;
        if (sw.n == 0) continue;
        const int n = T * sw.n;
        float* bil = static_cast<float*>(c->scratch_buf("ds_bil", sizeof(float) * (size_t)n * 6 * E * E));
        float* res = static_cast<float*>(c->scratch_buf("ds_out", sizeof(float) * (size_t)n * 6 * E * E));
        if (!bil || !res) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
        if (c->ds_half()) {
            const size_t ub = (size_t)n * 2 * Ep * Ep * 16;
            B16 x16{static_cast<uint4*>(c->scratch_buf("ds16_inh", ub)), static_cast<uint4*>(c->scratch_buf("ds16_inl", ub))};
            if (!x16.hi || !x16.lo) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
            { KTimer kt(c, "dsen2_gather", s);
              const dim3 gg((Ep * Ep + 255) / 256, sw.n, T);
              const int m = c->ds_blk_mode();
              if (m == 1) hipLaunchKernelGGL((k_sr_gather_b16<1>), gg, dim3(256), 0, s, d_s2, X, Y, sw, ws, cs, x16.hi, x16.lo);
              else hipLaunchKernelGGL((k_sr_gather_b16<0>), gg, dim3(256), 0, s, d_s2, X, Y, sw, ws, cs, x16.hi, x16.lo);
              hipLaunchKernelGGL(k_sr_bil_tile, dim3((E * E + 255) / 256, sw.n, T), dim3(256), 0, s, d_s2, X, Y, sw, ws, cs, bil);
              TTC_HIP(c, hipGetLastError()); }
            { const int m = c->ds_blk_mode();
              TTC_CHECK(m == 1 ? dsen2_core_h16<1>(c, x16, bil, n, E, E, res, s) : dsen2_core_h16<0>(c, x16, bil, n, E, E, res, s)); }
        } else {
            float* xin = static_cast<float*>(c->scratch_buf("ds_in", sizeof(float) * (size_t)n * 10 * Ep * Ep));
            if (!xin) return c->fail(TTC_ERR_NOMEM, "DSen2 scratch");
            { KTimer kt(c, "dsen2_gather", s);
              hipLaunchKernelGGL(k_sr_gather, dim3((Ep * Ep + 255) / 256, sw.n, T), dim3(256), 0, s, d_s2, X, Y, sw, ws, cs, xin);
              TTC_HIP(c, hipGetLastError()); }
            hipLaunchKernelGGL(k_sr_bil, dim3((E * E + 255) / 256, n), dim3(256), 0, s, xin, E, bil);
            TTC_HIP(c, hipGetLastError());
            TTC_CHECK(dsen2_core(c, xin, bil, n, E, E, res, s));
        }
        { KTimer kt(c, "dsen2_scatter", s);
          if (pass == 0) {
              for (int g = 0; g < 4; ++g)
                  if (grp[g + 1] > grp[g])
                      hipLaunchKernelGGL(k_sr_scatter, dim3((ws * ws + 255) / 256, grp[g + 1] - grp[g], T), dim3(256), 0, s,
                                         res, X, Y, sw, ws, cs, grp[g], grp[g + 1], d_s2);
          } else {
              hipLaunchKernelGGL(k_sr_scatter, dim3((ws * ws + 255) / 256, sw.n, T), dim3(256), 0, s, res, X, Y, sw, ws, cs, 0,
                                 sw.n, d_s2);
          }
          TTC_HIP(c, hipGetLastError()); }
    }
    return TTC_OK;
}

ttc_status decode_upsample_u16(ttc_ctx* c, const uint16_t* d10, const uint16_t* d20, int T, int h, int w, const AdjustMap* am10, float* d_out,
                               hipStream_t s) {
    if (!d10 || !d20 || !d_out || T < 1) return c->fail(TTC_ERR_ARG, "decode_upsample: bad argument");
    KTimer kt(c, "upsample_20m", s);
    const int P = 4 * h * w;
    const AdjustMap am = am10 ? *am10 : AdjustMap{2 * h, 2 * w, 0, 0};
    hipLaunchKernelGGL(k_decode_upsample, dim3((P + 255) / 256, T), dim3(256), 0, s, d10, d20, h, w, h % 2, w % 2, am, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}

ttc_status upsample_20m(ttc_ctx* c, const float* d10, const float* d20, int T, int h, int w, float* d_out, hipStream_t s) {
    if (!d10 || !d20 || !d_out || T < 1) return c->fail(TTC_ERR_ARG, "upsample_20m: bad argument");
    const int oy = h % 2, ox = w % 2, hh = (h - oy) / 2, ww = (w - ox) / 2;
    float* m = static_cast<float*>(c->scratch_buf("up_mean", sizeof(float) * (size_t)T * 2 * hh * ww));
    if (!m) return c->fail(TTC_ERR_NOMEM, "upsample scratch");
    KTimer kt(c, "upsample_20m", s);
    const int P = 4 * h * w;
    hipLaunchKernelGGL(k_upsample_20m, dim3((P + 255) / 256, T), dim3(256), 0, s, d10, d20, h, w, d_out);
    hipLaunchKernelGGL(k_mean2x2, dim3((hh * ww + 255) / 256, T), dim3(256), 0, s, d20, h, w, oy, ox, m);
    hipLaunchKernelGGL(k_upsample_40m, dim3((P + 255) / 256, T), dim3(256), 0, s, m, d20, h, w, oy, ox, d_out);
    TTC_HIP(c, hipGetLastError());
    return TTC_OK;
}
