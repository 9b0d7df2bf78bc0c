// Winograd F(2x2, 3x3) form of the 3x3 convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32) for the GroupNorm
// layers of the fp32 engine (ConvGRU gates / candidate, the conv_swish_gn blocks: src/train/src/model.py:251, :276, :416-442).
//
// Why: the direct implicit GEMM (conv3x3_mfma.hip) issues 9 multiply-accumulates per (output pixel, cin, cout) and sits at
// 75 % of the fp32 MFMA peak -- at ANY efficiency it cannot get below 0.76 ms for the gates launch.  Winograd's minimal
// filtering needs 16 / 4 = 4: 2.25 x fewer matrix instructions for the same sum (the transforms are additions).
//
//     Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A          d: 4 x 4 input patch, g: 3 x 3 kernel, Y: 2 x 2 outputs
//
// i.e. 16 independent GEMMs  M_xi[cout][tile] = sum_cin U_xi[cout][cin] * V_xi[cin][tile]  (xi = 4 a + b, the position in the
// transformed 4 x 4 patch), U = G g G^T precomputed on the host in double, V = B^T d B computed per workgroup in LDS.
//
// Mapping.  A workgroup (4 waves) owns 32 * TB tiles (a region of 8 x 4 TB tiles = 16 x 8 TB output pixels of one window) x
// 32 * NCB output channels, NCB * TB == 2.  The 16 xi are split between TWO waves (wave & 1: columns b in {0, 1} | {2, 3} of
// the transformed patch): 8 xi x one 32 x 32 accumulator tile = 128 accumulator registers per wave -> two workgroups per CU,
// so one workgroup's prologue / epilogue runs under the other's MFMAs exactly as in the direct kernel.  wave >> 1 selects the
// cout block (NCB == 2) or the tile block (TB == 2).
// Per 8-channel chunk: inputs [8][IR][18] are staged global -> registers -> LDS (issued one chunk ahead), every thread transforms
// TB patches (4 x 4 floats -> 16 values, 32 additions) into V[xi][k-half][tile][4 k-steps] -- written as consecutive floats by
// consecutive lanes, read back as ONE ds_read_b128 per (xi, lane) = the B operands of four k-steps.  The A operands (U) are
// not staged at all: they are read straight from global memory (a layer's U is <= 1.2 MB and lives in L2) as one 16-byte load
// per (xi, lane) and chunk through a 4-deep register ring.  Back-to-back MFMAs on the same accumulator run at full rate, so a
// wave's inner loop is xi-major: 8 x (1 ds_read_b128 + 1 global_load_dwordx4 + 4 v_mfma).
// Output transform: the row half (A^T M) is local to a wave; the column half needs both waves of a pair -- each sends the two
// partial sums of the OTHER output row through LDS (32 KB, the V buffers are free by then) and finishes its own row: wave & 1
// = row of the 2 x 2 output tile.  Epilogue ops (sSE gate, partial-conv ratio + swish), the deterministic GroupNorm partial
// sums and the flat [cout][y * Wp + x] output layout are those of conv_common.h's conv_epilogue_flat, so every consumer is
// unchanged.  fp32 throughout; the transforms' constants are 0, +-1, +-1/2: max |delta| vs the direct kernel ~1e-6 relative.
#include <algorithm>

#include "conv_common.h"

using namespace ttcconv;

namespace {

constexpr int kWCK = 8;          // input channels per chunk (4 k-steps of the 32x32x2 MFMA)
constexpr int kIC = 18;          // staged columns: 8 tiles * 2 + 2
constexpr int kRTX = 8;          // tiles per region along x

// workgroup barrier that waits for this wave's LDS traffic only: global loads (the next chunk's inputs, the A-operand ring) stay in
// flight across it
__device__ __forceinline__ void wbarrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NCB, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_wino(ConvArgs a, const float* __restrict__ Uall, long u_set_stride, int nchunk, int nks_last,
                                                        int RXn, int RYn, int ncp) {
    constexpr int TB = 2 / NCB;                  // tile blocks (of 32 tiles) per workgroup
    constexpr int NT = 32 * TB;                  // tiles per workgroup
    constexpr int RTY = 4 * TB;                  // tile rows of the region
    constexpr int IR = 2 * RTY + 2;              // staged rows
    constexpr int INE = kWCK * IR * kIC;         // floats of one staged chunk
    constexpr int NE = (INE + 255) / 256;        // staging elements per thread
    constexpr int VBUF = 16 * 2 * NT * 4;        // floats of one V buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_tile = smem;                       // [8][IR][18]
    float* Vb = smem + ((INE + 3) & ~3);         // [2][16 xi][2 k-half][NT][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xh = wave & 1, oth = wave >> 1;
    const int cbw = NCB == 2 ? oth : 0, tbw = NCB == 2 ? 0 : oth;
    const int tcol = lane & 31, hsel = lane >> 5;

    // XCD-aware work order (conv_common.h tile_index): consecutive ids of one XCD = neighbouring regions
    int lid;
    {
        const int G = gridDim.x, id = blockIdx.x;
        const int per = G >> 3, rem = G & 7, xcd = id & 7, slot = id >> 3;
        lid = xcd * per + (xcd < rem ? xcd : rem) + slot;
    }
    const int rx = lid % RXn;
    int rest = lid / RXn;
    const int cp = rest % ncp; rest /= ncp;
    const int ry = rest % RYn;
    const int n = rest / RYn;
    const int set = n / a.n_per_set, nn = n - set * a.n_per_set;

    const int Wp = a.Wp, Hp = a.Hp, H = Hp - 2, W = Wp - 2;
    const int plane = Hp * Wp;
    const float* seg0 = a.seg[0].base + (long)nn * a.seg[0].stride_n + a.seg[0].set_off[set];
    const float* seg1 = a.seg[1].C > 0 ? a.seg[1].base + (long)nn * a.seg[1].stride_n + a.seg[1].set_off[set] : seg0;
    const int C0 = a.seg[0].C, Cin = a.Cin;
    const float* aux = a.aux ? a.aux + (long)set * a.aux_set_stride : nullptr;
    // U: [set][cout block][chunk][xi][k-half][cout 32][4 k-steps]
    const float4* Uw = reinterpret_cast<const float4*>(Uall + (long)set * u_set_stride) + ((long)(cp * NCB + cbw) * nchunk) * (16 * 2 * 32)
                       + hsel * 32 + tcol;

    // ---- staging geometry: element e = tid + 256 k of the [8][IR][18] chunk image
    const int y0 = ry * (2 * RTY), x0 = rx * (2 * kRTX);
    int goff[NE];                                 // offset inside the plane (clamped into it) | channel << 24
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        int e = tid + 256 * k;
        e = e < INE ? e : INE - 1;
        const int cl = e / (IR * kIC), r2 = e - cl * (IR * kIC);
        const int row = r2 / kIC, col = r2 - row * kIC;
        const int yy = min(y0 + row, Hp - 1), xx = min(x0 + col, Wp - 1);
        goff[k] = (yy * Wp + xx) | (cl << 24);
    }
    float g[NE];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            int ci = c * kWCK + (goff[k] >> 24);
            ci = ci < Cin ? ci : Cin - 1;         // pad channels meet zero weights: any finite plane will do
            const float* src = ci < C0 ? seg0 + (long)ci * plane : seg1 + (long)(ci - C0) * plane;
            g[k] = src[goff[k] & 0xffffff];
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + 256 * k;
            if (e < INE) in_tile[e] = g[k];
        }
    };
    // ---- input transform: thread -> (k-step j, tile, k-half hs) of each tile block: channel 2 j + hs of the chunk
    const int tj = lane & 3, ttl = (wave & 1) * 16 + (lane >> 2), ths = wave >> 1;
    auto transform = [&](int buf) {
        float* V = Vb + buf * VBUF;
#pragma unroll
        for (int p = 0; p < TB; ++p) {
            const int tile = p * 32 + ttl;
            const int tyl = tile >> 3, txl = tile & 7;
            const float* src = in_tile + ((2 * tj + ths) * IR + 2 * tyl) * kIC + 2 * txl;
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo2 = *reinterpret_cast<const float2*>(src + r * kIC);
                const float2 hi2 = *reinterpret_cast<const float2*>(src + r * kIC + 2);
                d[r][0] = lo2.x; d[r][1] = lo2.y; d[r][2] = hi2.x; d[r][3] = hi2.y;
            }
            float t[4][4];                        // B^T d
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                t[0][cidx] = d[0][cidx] - d[2][cidx];
                t[1][cidx] = d[1][cidx] + d[2][cidx];
                t[2][cidx] = d[2][cidx] - d[1][cidx];
                t[3][cidx] = d[1][cidx] - d[3][cidx];
            }
            float* dst = V + (ths * NT + tile) * 4 + tj;
#pragma unroll
            for (int r = 0; r < 4; ++r) {         // (B^T d) B
                dst[((r * 4 + 0) * 2) * NT * 4] = t[r][0] - t[r][2];
                dst[((r * 4 + 1) * 2) * NT * 4] = t[r][1] + t[r][2];
                dst[((r * 4 + 2) * 2) * NT * 4] = t[r][2] - t[r][1];
                dst[((r * 4 + 3) * 2) * NT * 4] = t[r][1] - t[r][3];
            }
        }
    };

    f32x16 acc[8];                                // xi_l = 2 a + bl  <->  xi = 4 a + 2 xh + bl
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // A-operand ring: slot i & 3 holds (chunk, xi_l = i & 7) of the running sequence i = 8 c + xi_l; loads run 3 ahead
    float4 aring[4];
    const int total = 8 * nchunk;
    auto a_load = [&](int i) {
        const int c = i >> 3, xl = i & 7;
        const int xi = 4 * (xl >> 1) + 2 * xh + (xl & 1);
        return Uw[((long)c * 16 + xi) * (2 * 32)];
    };
#pragma unroll
    for (int i = 0; i < 3; ++i) aring[i] = a_load(i < total ? i : total - 1);

    // ---- prologue
    stage_load(0);
    stage_store();
    __syncthreads();
    if (nchunk > 1) stage_load(1);
    transform(0);
    __syncthreads();
    if (nchunk > 1) stage_store();

    const float4* Vr = reinterpret_cast<const float4*>(Vb) + (hsel * NT + tbw * 32 + tcol);
    for (int c = 0; c < nchunk; ++c) {
        wbarrier();                               // V[c & 1] is complete, in_tile holds chunk c + 1
        const bool more2 = c + 2 < nchunk, more1 = c + 1 < nchunk;
        if (more2) stage_load(c + 2);
        if (more1) transform((c + 1) & 1);
        const float4* Vc = Vr + (c & 1) * (VBUF / 4);
        const int nks = (c == nchunk - 1) ? nks_last : 4;
        auto mfma_xi = [&](int xl) {
            const int i = 8 * c + xl;
            const int xi = 4 * (xl >> 1) + 2 * xh + (xl & 1);
            const float4 b = Vc[xi * (2 * NT)];
            const float4 av = aring[xl & 3];
            const int inext = i + 3;
            aring[(xl + 3) & 3] = a_load(inext < total ? inext : total - 1);
            acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b.x, acc[xl], 0, 0, 0);
            if (nks > 1) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b.y, acc[xl], 0, 0, 0);
            if (nks > 2) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b.z, acc[xl], 0, 0, 0);
            if (nks > 3) acc[xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b.w, acc[xl], 0, 0, 0);
        };
#pragma unroll
        for (int xl = 0; xl < 4; ++xl) mfma_xi(xl);
        wbarrier();                               // every transform of chunk c + 1 has read in_tile
        if (more2) stage_store();
#pragma unroll
        for (int xl = 4; xl < 8; ++xl) mfma_xi(xl);
    }

    // ---- output transform.  acc[2 a + bl] = M[a][b = 2 xh + bl]
    //   T0[b] = M0b + M1b + M2b, T1[b] = M1b - M2b - M3b;  Y[i][0] = Ti0 + Ti1 + Ti2, Y[i][1] = Ti1 - Ti2 - Ti3
    //   xh = 0 holds b = 0, 1: p[i] = (Ti0 + Ti1, Ti1);  xh = 1 holds b = 2, 3: p[i] = (Ti2, -Ti2 - Ti3);  Y[i] = p0[i] + p1[i]
    float keep[16][2], give[16][2];               // own output row (i = xh) | the other row, for the partner wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float T[2][2];
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            T[0][bl] = acc[0 + bl][r] + acc[2 + bl][r] + acc[4 + bl][r];
            T[1][bl] = acc[2 + bl][r] - acc[4 + bl][r] - acc[6 + bl][r];
        }
        float p[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            p[i][0] = xh == 0 ? T[i][0] + T[i][1] : T[i][0];
            p[i][1] = xh == 0 ? T[i][1] : -T[i][0] - T[i][1];
        }
        keep[r][0] = xh == 0 ? p[0][0] : p[1][0]; keep[r][1] = xh == 0 ? p[0][1] : p[1][1];
        give[r][0] = xh == 0 ? p[1][0] : p[0][0]; give[r][1] = xh == 0 ? p[1][1] : p[0][1];
    }
    __syncthreads();                              // the V buffers are free
    float* ex = Vb + (oth * 2) * (16 * 2 * 64);   // [pair][sender xh][r][jj][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ex[xh * (16 * 2 * 64) + (r * 2 + 0) * 64 + lane] = give[r][0];
        ex[xh * (16 * 2 * 64) + (r * 2 + 1) * 64 + lane] = give[r][1];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        keep[r][0] += ex[(xh ^ 1) * (16 * 2 * 64) + (r * 2 + 0) * 64 + lane];
        keep[r][1] += ex[(xh ^ 1) * (16 * 2 * 64) + (r * 2 + 1) * 64 + lane];
    }

    // ---- epilogue op, GroupNorm partial sums, stores.  This wave: output row 2 ty + xh, columns 2 tx, 2 tx + 1 of its 32 tiles
    const int tile = tbw * 32 + tcol;
    const int ty = ry * RTY + (tile >> 3), tx = rx * kRTX + (tile & 7);
    const int y = 2 * ty + xh, xa = 2 * tx;
    const bool vrow = y < H;
    const bool v0 = vrow && xa < W, v1 = vrow && xa + 1 < W;
    const int cbase = (cp * NCB + cbw) * 32;
    if (EPI == EPI_SSE) {
        float dot0 = 0.f, dot1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float k1 = aux[(r & 3) + 8 * (r >> 2) + 4 * hsel];
            dot0 += k1 * keep[r][0]; dot1 += k1 * keep[r][1];
        }
        dot0 += __shfl_xor(dot0, 32); dot1 += __shfl_xor(dot1, 32);
        const float g0 = sigmoidf_(dot0), g1 = sigmoidf_(dot1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { keep[r][0] *= g0; keep[r][1] *= g1; }
    }
    if (EPI == EPI_SWISH) {
        float r0 = 1.0f, r1 = 1.0f;
        if (a.same_pad) {
            const bool ey = (y == 0) || (y == H - 1);
            const bool ex0 = (xa == 0) || (xa == W - 1), ex1 = (xa + 1 == W - 1);
            r0 = (ey && ex0) ? 2.25f : ((ey || ex0) ? 1.5f : 1.0f);
            r1 = (ey && ex1) ? 2.25f : ((ey || ex1) ? 1.5f : 1.0f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float u0 = keep[r][0] * r0, u1 = keep[r][1] * r1;
            keep[r][0] = u0 * sigmoidf_(u0); keep[r][1] = u1 * sigmoidf_(u1);
        }
    }
    if (a.stats) {
        float red[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float u0 = keep[4 * k + rr][0], u1 = keep[4 * k + rr][1];
                if (v0) { s += u0; q += u0 * u0; }
                if (v1) { s += u1; q += u1 * u1; }
            }
            red[2 * k] = s; red[2 * k + 1] = q;
        }
        half_wave_sums(red);
        if (tcol == 31) {
            constexpr int SW = 4 / NCB;            // waves that contribute to one cout block of a region
            const long slots = (long)RXn * RYn * SW;
            const int slot = (ry * RXn + rx) * SW + (NCB == 2 ? xh : wave);
            float2* base = reinterpret_cast<float2*>(a.stats) + (long)n * (a.Cout / 4) * slots + slot;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int quad = cbase / 4 + 2 * k + hsel;
                if (quad * 4 < a.Cout) base[quad * slots] = make_float2(red[2 * k], red[2 * k + 1]);
            }
        }
    }
    float* outn = a.out + (long)n * a.out_stride_n;
    const long opix = (long)y * Wp + xa;
    const bool vec = (((a.out_plane | (long)Wp) & 1L) == 0) && v1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * hsel;
        if (co >= a.Cout) continue;
        float* o = outn + (long)co * a.out_plane + opix;
        if (vec) *reinterpret_cast<float2*>(o) = make_float2(keep[r][0], keep[r][1]);
        else {
            if (v0) o[0] = keep[r][0];
            if (v1) o[1] = keep[r][1];
        }
    }
}

template <int NCB, int EPI>
hipError_t launch_w(const ConvArgs& a, const PackedConv& pw, int n, hipStream_t s) {
    constexpr int TB = 2 / NCB, NT = 32 * TB, IR = 2 * 4 * TB + 2;
    const size_t lds = (size_t)(((kWCK * IR * kIC + 3) & ~3) + 2 * 16 * 2 * NT * 4) * sizeof(float);
    static LdsConfig lds_cfg;
    if (hipError_t e = lds_cfg.ensure(&conv3x3_wino<NCB, EPI>, lds); e != hipSuccess) return e;
    const int H = a.Hp - 2, W = a.Wp - 2;
    const int TX = (W + 1) / 2, TY = (H + 1) / 2;
    const int RXn = (TX + kRTX - 1) / kRTX, RYn = (TY + 4 * TB - 1) / (4 * TB);
    const int ncp = (a.Cout + 32 * NCB - 1) / (32 * NCB);
    if ((long)a.Hp * a.Wp >= (1L << 24)) return hipErrorInvalidValue;
    const int rem = a.Cin - kWCK * (pw.nchunk_w - 1);
    const int nks_last = (rem + 1) / 2;
    hipLaunchKernelGGL((conv3x3_wino<NCB, EPI>), dim3((unsigned)((long)RXn * RYn * ncp * n)), dim3(256), lds, s, a, pw.d_wu, pw.set_stride_w,
                       pw.nchunk_w, nks_last, RXn, RYn, ncp);
    return hipGetLastError();
}

}  // namespace

// GroupNorm partial-sum slots per (window, channel quad) of the Winograd kernels: one per region and contributing wave
int conv_wino_stat_slots(int Hp, int Wp, int Cout) {
    const int H = Hp - 2, W = Wp - 2;
    const int TX = (W + 1) / 2, TY = (H + 1) / 2;
    const int ncb = Cout >= 64 ? 2 : 1, tb = 2 / ncb;
    return ((TX + kRTX - 1) / kRTX) * ((TY + 4 * tb - 1) / (4 * tb)) * (4 / ncb);
}

// U = G g G^T per (cin, cout) in double, packed [set][cout block 32][chunk][xi 16][k-half 2][cout 32][k-step 4]:
// channel of (chunk c, k-half h, k-step s) = 8 c + 2 s + h.  Returns floats per set.
long conv_pack_wino(const float* const* hwio, int nsets, int Cin, int Cout, std::vector<float>& out, int* nchunk_out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nchunk = (Cin + kWCK - 1) / kWCK, ncb = (Cout + 31) / 32;
    const long per_set = (long)ncb * nchunk * 16 * 2 * 32 * 4;
    out.assign((size_t)per_set * nsets, 0.0f);
    for (int s = 0; s < nsets; ++s)
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci) {
                double g[3][3], t[4][3];
                for (int u = 0; u < 3; ++u)
                    for (int v = 0; v < 3; ++v) g[u][v] = hwio[s][((long)(u * 3 + v) * Cin + ci) * Cout + co];
                for (int aa = 0; aa < 4; ++aa)
                    for (int v = 0; v < 3; ++v) t[aa][v] = G[aa][0] * g[0][v] + G[aa][1] * g[1][v] + G[aa][2] * g[2][v];
                const int cb = co >> 5, col = co & 31, c = ci / kWCK, k = ci % kWCK, h = k & 1, st = k >> 1;
                for (int aa = 0; aa < 4; ++aa)
                    for (int bb = 0; bb < 4; ++bb) {
                        const double u = t[aa][0] * G[bb][0] + t[aa][1] * G[bb][1] + t[aa][2] * G[bb][2];
                        const int xi = aa * 4 + bb;
                        out[(size_t)s * per_set + (((((long)cb * nchunk + c) * 16 + xi) * 2 + h) * 32 + col) * 4 + st] = (float)u;
                    }
            }
    if (nchunk_out) *nchunk_out = nchunk;
    return per_set;
}

hipError_t conv_launch_wino(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s) {
    if (!pw.d_wu) return hipErrorInvalidValue;
    const bool two = pw.Cout >= 64;
    if (two && epi == EPI_RAW) return launch_w<2, EPI_RAW>(a, pw, n, s);
    if (two && epi == EPI_SWISH) return launch_w<2, EPI_SWISH>(a, pw, n, s);
    if (!two && epi == EPI_SSE) return launch_w<1, EPI_SSE>(a, pw, n, s);
    if (!two && epi == EPI_SWISH) return launch_w<1, EPI_SWISH>(a, pw, n, s);
    return hipErrorInvalidValue;
}
