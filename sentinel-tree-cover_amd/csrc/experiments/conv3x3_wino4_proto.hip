// PROTOTYPE, compile-only (end of round 4; never run: no GPU minutes were left).  Winograd F(4x4, 3x3) on v_mfma_f32_16x16x4_f32, ONE WAVE
// OWNS WHOLE TILES -- the design of docs/NEXT_winograd_f4.md, written down far enough for hipcc to answer the questions that decide it:
// do 36 accumulator quads + an operand ring + the 6x6 transforms fit 256 VGPRs without spills, and what does the instruction mix look like?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I.. -c conv3x3_wino4_proto.hip -Rpass-analysis=kernel-resource-usage
// Not part of the library (the Makefile globs csrc/*.hip only).  Simplifications against the product kernel: one tile per loop iteration with
// its own start-up (no cross-tile chunk stream), EPI_RAW + GroupNorm sums only, Cout a multiple of 64, both pitches even, W % 4 == 0.
//
// Mapping.  Workgroup = 4 waves = 4 cout blocks of 16 over the SAME 16 tiles (a 4 x 4-tile region = 16 x 16 output pixels of one window).
//   acc[xi] (xi = 6 a + b, 36 of them) is one 16 x 16 D tile: lane l holds column n = l & 15 (tile), rows 4 (l >> 4) + r (couts), r = 0..3.
//   A operand (U = G g G^T, global memory): lane l -> cout l & 15, k = l >> 4; float2 = the two k-steps of an 8-channel chunk.
//   B operand (V = B^T d B, LDS):           lane l -> tile l & 15, k = l >> 4; float2 likewise: V[xi][k 4][tile 16][s 2], lane-linear.
// Per chunk and wave: 36 x (ds_read_b64 + global_load_dwordx2 + 2 MFMA) = 72 MFMAs x 32 cycles.
#include "../conv_common.h"

using namespace ttcconv;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCK = 8;                 // input channels per chunk = 2 k-steps of 4
constexpr int kIR = 18;                // staged rows / columns: 4 tiles x 4 + 2
constexpr int kIP = 20;                // staged row pitch (floats): even, so that a patch row is three aligned 8-byte reads
constexpr int kINE = kCK * kIR * kIP;  // floats of one staged chunk image
constexpr int kNE = (kCK * kIR * (kIR / 2) + 255) / 256;   // staging float2 per thread (6)
constexpr int kVB = 36 * 4 * 16 * 2;   // floats of one V buffer (18 KB)
constexpr int kRING = 8;              // A operands requested ahead (xi)

struct Wino4Args {
    ConvArgs a;
    const float* U; long u_set_stride;       // [set][cout block 16][chunk][xi 36][k 4][cout 16][s 2]
    int nchunk, RXn, RYn, ncq, ntiles;
    unsigned long long m_rx, m_cq, m_ry, m_set;
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256, 2) void conv3x3_wino4(Wino4Args wa) {
    typedef const __attribute__((address_space(4))) Wino4Args* KArgs;
    const KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    auto args = [&]() { KArgs q = kp; asm volatile("" : "+s"(q)); return q; };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_tile = smem;                       // [8][18][20]
    float* Vb = smem + kINE;                     // [2][36][4][16][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = args()->a.Wp, Hp = args()->a.Hp, H = Hp - 2, W = Wp - 2, plane = Hp * Wp;
    const int T = args()->nchunk, RXn = args()->RXn, RYn = args()->RYn;
    const int P = gridDim.x, xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    const int nx = (P >> 3) + (xcd < (P & 7) ? 1 : 0);
    const int ntiles = args()->ntiles, per = ntiles >> 3, rem = ntiles & 7;
    const int tcnt = per + (xcd < rem ? 1 : 0), tstart = xcd * per + (xcd < rem ? xcd : rem);
    auto mdiv = [](int x, unsigned long long m) { return (int)(((unsigned long long)(unsigned)x * m) >> 40); };

    // staging pair k of this thread inside the [8][18][20] image: row | col << 8 | channel << 24
    int rc[kNE];
#pragma unroll
    for (int k = 0; k < kNE; ++k) {
        int e = tid + 256 * k;
        e = e < kCK * kIR * (kIR / 2) ? e : kCK * kIR * (kIR / 2) - 1;
        const int cl = e / (kIR * (kIR / 2)), r2 = e - cl * (kIR * (kIR / 2));
        const int row = r2 / (kIR / 2), col = 2 * (r2 - row * (kIR / 2));
        rc[k] = row | (col << 8) | (cl << 24);
    }
    // input transform: thread -> (patch p = tid & 127 = tile * 8 + channel, half = wave >> 1 (wave-uniform: no divergent paths): rows 3 half .. 3 half + 2 of B^T d)
    const int tp = tid & 127, thalf = wave >> 1, tch = tp & 7, ttile = tp >> 3;
    const float* tsrc = in_tile + (tch * kIR + 4 * (ttile >> 2)) * kIP + 4 * (ttile & 3);
    // V element (xi, k = ch & 3, tile, s = ch >> 2)
    float* tdst = Vb + ((tch & 3) * 16 + ttile) * 2 + (tch >> 2);

    const float2* Vr = reinterpret_cast<const float2*>(Vb) + lane;
    f32x4 acc[36];

    for (int tk = wslot; tk < tcnt; tk += nx) {
        // ---- tile id -> (window, region row, cout quad, region column)
        const KArgs ka = args();
        const int lid = tstart + tk;
        int rest = mdiv(lid, ka->m_rx);
        const int rx = lid - rest * RXn;
        int q = mdiv(rest, ka->m_cq);
        const int cq = rest - q * ka->ncq; rest = q;
        q = mdiv(rest, ka->m_ry);
        const int ry = rest - q * RYn, n = q;
        const int set = mdiv(n, ka->m_set), nn = n - set * ka->a.n_per_set;
        const float* seg0 = ka->a.seg[0].base + (long)nn * ka->a.seg[0].stride_n + ka->a.seg[0].set_off[set];
        const float* seg1 = ka->a.seg[1].C > 0 ? ka->a.seg[1].base + (long)nn * ka->a.seg[1].stride_n + ka->a.seg[1].set_off[set] : seg0;
        const int Cin = ka->a.Cin, C0 = ka->a.seg[0].C;
        const float2* uw = reinterpret_cast<const float2*>(ka->U + (long)set * ka->u_set_stride) + ((long)(cq * 4 + wave) * T) * (36 * 64) + lane;
        const int y0 = ry * 16, x0 = rx * 16;
        int goff[kNE];
#pragma unroll
        for (int k = 0; k < kNE; ++k) {
            const int yy = min(y0 + (rc[k] & 0xff), Hp - 1), xx = min(x0 + ((rc[k] >> 8) & 0xff), Wp - 2);
            goff[k] = yy * Wp + xx;
        }
        float2 g[kNE];
        auto stage_load = [&](int c) {
#pragma unroll
            for (int k = 0; k < kNE; ++k) {
                int ci = c * kCK + (rc[k] >> 24);
                ci = ci < Cin ? ci : Cin - 1;
                const bool lo = ci < C0;
                const float* base = lo ? seg0 : seg1;
                const unsigned off = __umul24((unsigned)(lo ? ci : ci - C0), (unsigned)plane) + (unsigned)goff[k];
                g[k] = *reinterpret_cast<const float2*>(base + off);
            }
        };
        auto stage_store = [&]() {
#pragma unroll
            for (int k = 0; k < kNE; ++k) {
                const int e = tid + 256 * k;
                if (e < kCK * kIR * (kIR / 2)) {
                    const int cl = rc[k] >> 24, row = rc[k] & 0xff, col = (rc[k] >> 8) & 0xff;
                    *reinterpret_cast<float2*>(in_tile + (cl * kIR + row) * kIP + col) = g[k];
                }
            }
        };
        // B^T d B of half a patch: three rows of T = B^T d (from five input rows), then the column pass row by row
        auto transform = [&](int buf) {
            float* dst = tdst + buf * kVB;
            float t[3][6];
#pragma unroll
            for (int c2 = 0; c2 < 3; ++c2) {         // two columns at a time: ten live inputs instead of thirty
                float2 d[5];
#pragma unroll
                for (int r = 0; r < 5; ++r) d[r] = *reinterpret_cast<const float2*>(tsrc + (r + thalf) * kIP + 2 * c2);
                if (thalf == 0) {        // rows 0, 1, 2 of B^T from input rows 0 .. 4
                    t[0][2 * c2] = 4.f * d[0].x - 5.f * d[2].x + d[4].x;            t[0][2 * c2 + 1] = 4.f * d[0].y - 5.f * d[2].y + d[4].y;
                    t[1][2 * c2] = -4.f * (d[1].x + d[2].x) + d[3].x + d[4].x;      t[1][2 * c2 + 1] = -4.f * (d[1].y + d[2].y) + d[3].y + d[4].y;
                    t[2][2 * c2] = 4.f * (d[1].x - d[2].x) - d[3].x + d[4].x;       t[2][2 * c2 + 1] = 4.f * (d[1].y - d[2].y) - d[3].y + d[4].y;
                } else {                 // rows 3, 4, 5 from input rows 1 .. 5 (d[r] holds input row r + 1)
                    t[0][2 * c2] = -2.f * d[0].x - d[1].x + 2.f * d[2].x + d[3].x;  t[0][2 * c2 + 1] = -2.f * d[0].y - d[1].y + 2.f * d[2].y + d[3].y;
                    t[1][2 * c2] = 2.f * d[0].x - d[1].x - 2.f * d[2].x + d[3].x;   t[1][2 * c2 + 1] = 2.f * d[0].y - d[1].y - 2.f * d[2].y + d[3].y;
                    t[2][2 * c2] = 4.f * d[0].x - 5.f * d[2].x + d[4].x;            t[2][2 * c2 + 1] = 4.f * d[0].y - 5.f * d[2].y + d[4].y;
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float* tr = t[a];
                float v[6];
                v[0] = 4.f * tr[0] - 5.f * tr[2] + tr[4];
                v[1] = -4.f * (tr[1] + tr[2]) + tr[3] + tr[4];
                v[2] = 4.f * (tr[1] - tr[2]) - tr[3] + tr[4];
                v[3] = -2.f * tr[1] - tr[2] + 2.f * tr[3] + tr[4];
                v[4] = 2.f * tr[1] - tr[2] - 2.f * tr[3] + tr[4];
                v[5] = 4.f * tr[1] - 5.f * tr[3] + tr[5];
#pragma unroll
                for (int b = 0; b < 6; ++b) dst[((3 * thalf + a) * 6 + b) * (4 * 16 * 2)] = v[b];
            }
        };

        // ---- start-up of the tile
        float2 A[kRING];
        auto a_load = [&](int c, int xi) { return uw[((long)c * 36 + xi) * 64]; };
#pragma unroll
        for (int i = 0; i < kRING; ++i) A[i] = a_load(0, i);
        // stream position p of the tile = its chunk p; the staged image holds position c + 1 when chunk c starts (as in the product kernel)
        stage_load(0);
        __syncthreads();                           // the previous tile's last chunk has been read
        stage_store();
        __syncthreads();
        stage_load(1 < T ? 1 : 0);
        transform(0);
        __syncthreads();
        stage_store();
        stage_load(2 < T ? 2 : T - 1);
        int par = 0;
        for (int c = 0; c < T; ++c) {
            lds_barrier();                          // V[par] complete, the staged image holds stream position c + 1
            const float2* Vc = Vr + par * (kVB / 2);
            const int cn = c + 1 < T ? c + 1 : c;
            transform(par ^ 1);                     // interleaves with the first half's matrix instructions
            float2 b = Vc[0];
            auto half = [&](int x0) {
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    const int xi = x0 + k;
                    const float2 bn = Vc[(xi + 1 < 36 ? xi + 1 : xi) * 64];
                    const float2 a = A[xi % kRING];
                    if (c == 0) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, z, 0, 0, 0);
                    } else {
                        acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[xi], 0, 0, 0);
                    }
                    acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[xi], 0, 0, 0);
                    A[xi % kRING] = xi + kRING < 36 ? a_load(c, xi + kRING) : a_load(cn, xi + kRING - 36);
                    b = bn;
                }
            };
            half(0);
            lds_barrier();                          // every transform of stream position c + 1 has read the staged image
            stage_store();                          // stream position c + 2
            stage_load(c + 3 < T ? c + 3 : T - 1);
            __builtin_amdgcn_sched_barrier(0);
            half(18);
            par ^= 1;
        }

        // ---- epilogue, wave-local: A^T M A per cout, 16-px blocks leave as 8-byte stores, GroupNorm sums of the lane's channel group
        {
            const KArgs kb = args();
            float* outn = kb->a.out + (long)n * kb->a.out_stride_n;
            const long out_plane = kb->a.out_plane;
            const int tile = lane & 15, ty = 4 * ry + (tile >> 2), tx = 4 * rx + (tile & 3);
            const int yb = 4 * ty, xb = 4 * tx;
            const bool vcol = xb < W;               // W % 4 == 0: a tile's four columns are valid together
            float ssum = 0.f, ssq = 0.f;
            typedef float pair_f __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (cq * 4 + wave) * 16 + 4 * (lane >> 4) + r;
                float R[4][6];                       // rows of A^T M
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const float m0 = acc[b][r], m1 = acc[6 + b][r], m2 = acc[12 + b][r], m3 = acc[18 + b][r], m4 = acc[24 + b][r], m5 = acc[30 + b][r];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    R[0][b] = m0 + s12 + s34;
                    R[1][b] = d12 + 2.f * d34;
                    R[2][b] = s12 + 4.f * s34;
                    R[3][b] = d12 + 8.f * d34 + m5;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float s12 = R[i][1] + R[i][2], d12 = R[i][1] - R[i][2], s34 = R[i][3] + R[i][4], d34 = R[i][3] - R[i][4];
                    const float o0 = R[i][0] + s12 + s34, o1 = d12 + 2.f * d34, o2 = s12 + 4.f * s34, o3 = d12 + 8.f * d34 + R[i][5];
                    const bool ok = vcol && yb + i < H;
                    float sm = o0; float sq = o0 * o0;
                    sm += o1; sq += o1 * o1; sm += o2; sq += o2 * o2; sm += o3; sq += o3 * o3;
                    ssum += ok ? sm : 0.f; ssq += ok ? sq : 0.f;
                    if (ok) {
                        float* o = outn + (long)co * out_plane + (long)(yb + i) * Wp + xb;
                        pair_f p0, p1; p0.x = o0; p0.y = o1; p1.x = o2; p1.y = o3;
                        *reinterpret_cast<pair_f*>(o) = p0;
                        *reinterpret_cast<pair_f*>(o + 2) = p1;
                    }
                }
            }
            if (kb->a.stats) {                       // a lane's four couts are one GroupNorm quad: sum over the 16 lanes (tiles) of its DPP row
                float red[2] = {ssum, ssq};
#pragma unroll
                for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x111, 0xf>(red[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x112, 0xf>(red[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x114, 0xf>(red[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x118, 0xf>(red[i]);
                if ((lane & 15) == 15) {
                    const long slots = (long)RXn * RYn;
                    const int quad = (cq * 4 + wave) * 4 + (lane >> 4);
                    float2* base = reinterpret_cast<float2*>(kb->a.stats) + ((long)n * (kb->a.Cout / 4) + quad) * slots + ry * RXn + rx;
                    *base = make_float2(red[0], red[1]);
                }
            }
        }
    }
}

}  // namespace

// host side of the prototype: launch geometry only (the packer is conv_pack_wino with G 6 x 3 and the layout above)
hipError_t conv_launch_wino4_proto(const ConvArgs& a, const float* d_u, long u_set_stride, int nchunk, int n, int cus, hipStream_t s) {
    const int H = a.Hp - 2, W = a.Wp - 2;
    const int RXn = (W + 15) / 16, RYn = (H + 15) / 16, ncq = (a.Cout + 63) / 64;
    const long ntiles = (long)RXn * RYn * ncq * n;
    auto magic = [](long d) { return (1ULL << 40) / (unsigned long long)d + 1ULL; };
    if (ntiles >= (1L << 24) || (a.Wp & 1) || (W & 3) || (a.Cout & 63) || nchunk < 2) return hipErrorInvalidValue;
    Wino4Args wa{a, d_u, u_set_stride, nchunk, RXn, RYn, ncq, (int)ntiles, magic(RXn), magic(ncq), magic(RYn), magic(a.n_per_set)};
    const size_t lds = (size_t)(kINE + 2 * kVB) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv3x3_wino4, dim3((unsigned)std::min<long>(ntiles, 2L * cus)), dim3(256), lds, s, wa);
    return hipGetLastError();
}
