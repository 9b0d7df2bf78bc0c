// Implicit-GEMM 3x3 convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every tf.nn.convolution / Conv2D of the two reference graphs
// (src/train/src/model.py:251, :276, :416-442; superresolve_graph.pb Conv2D nodes).
//
// Formulation ("flattened padded plane"): the producer of an activation stores it planar
// and ALREADY padded, [n][Cin][Hp][Wp].  With q = y*Wp + x,
//     out[co][q] = sum_{ci,dy,dx} w[co][ci][dy][dx] * in[ci][q + dy*Wp + dx]
// so a tap is a linear offset, a workgroup owns BQ = 512 consecutive q of one window and
// stages [CK][BQ + 2*Wp + 2] input floats + [9][CK][BN] weights in LDS per Cin chunk.
// The two junk columns per row (x >= Wp-2) are computed and dropped: 1.2 % waste at
// Wp = 174, versus 11.6 % for 32-pixel row tiles at W = 172.
//
// GEMM view per MFMA (32x32x2, D = A*B + C):  A[i = lane&31][k = lane>>5] = weight of
// cout i, B[k][j = lane&31] = input of pixel j, D[row = cout][col = pixel]; a wave holds
// NCG x QG accumulator tiles (NCG cout-groups of 32, QG pixel-groups of 32).
// fp32 MFMA is an exact k-ordered fmaf chain, so results match an fp32 CPU reference to
// summation-order rounding.
//
// Schedule (measured on MI355X, ConvGRU gates 49->64, 72 x 172^2, 120.2 GFLOP per launch):
//   4 waves x (2 cout-groups x 4 pixel-groups) per workgroup, single LDS stage (57.6 KB), TWO
//   workgroups per CU: one workgroup's global->register->LDS staging, tile prologue and epilogue
//   store burst run under the other's MFMAs.                         1.04 ms = 115 TFLOP/s (73 %)
//   The stage's global loads are branch-free and all issued before the first use (with branches or
//   an early select hipcc waits vmcnt(0) after every load: 8 round trips per stage, 1.85 ms).
// Alternatives kept under csrc/experiments/ with their numbers:
//   - 8 waves, one workgroup per CU, double-buffered LDS + operand prefetch:       1.10 ms (70 %)
//   - the same made persistent (tile loop inside, epilogue under the next tile):   1.20 ms -- 109 SGPR
//     spills (v_readlane traffic in the MFMA loop) eat the gain.
//   Ablations of the 8-wave variant: no staging 1.02 ms; no staging + no output stores 0.99 ms; the bare
//   MFMA + ds_read loop therefore runs at ~84 % of the 157 TFLOP/s peak (2.34 GHz measured clock,
//   SQ_VALU_MFMA_BUSY 66-70 % overall), i.e. ~0.84 ms is the floor of this formulation.
#include <algorithm>

#include "conv_common.h"

using namespace ttcconv;

namespace {

template <int CK, int NCG, int EPI, bool TRACE = false>
__global__ __launch_bounds__(kThreads, (NCG == 1 && CK <= 8) ? 3 : 2) void conv3x3_f32(ConvArgs a, int nchunk, int nblk_q, int ncb) {
    constexpr int BN = NCG * 32;
    // TRACE instantiation (probe aid, env TTC_F32_TRACE): wave 0 stamps s_memtime at the phase boundaries of its tile
    unsigned long long* tr = (TRACE && threadIdx.x == 0) ? a.trace + (long)blockIdx.x * 64 : nullptr;
    if (TRACE && tr) tr[0] = __builtin_amdgcn_s_memtime();
    // Tried: s_setprio(3) outside the MFMA blocks, 0 inside.  The per-workgroup trace shows WHY the non-MFMA phases are slow (the
    // wave inside its MFMA block wins issue arbitration; kernel entry -> first chunk 22 k cycles, epilogue 43 k of a 272 k-cycle
    // tile) and that priority fixes exactly that (3.6 k / 32 k) -- but the MFMA blocks then share the pipe (24.6 k -> 42 k per
    // chunk) and the tile takes the same 272 k: two waves x 115 k cycles of MFMA issue per SIMD is 85 % of it either way.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Wp = a.Wp, Hp = a.Hp;
    const int plane = Hp * Wp;
    const int halo = 2 * Wp + 2;
    const int TL = kBQ + halo;
    const int TLp = (TL + 3) & ~3;
    float* in_tile = smem;                   // [CK][TLp]
    float* w_tile = smem + CK * TLp;         // [9][CK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    int bq, cb, n;
    tile_index(nblk_q, ncb, bq, cb, n);
    const int set = n / a.n_per_set, nn = n - set * a.n_per_set;
    const int q0 = bq * kBQ;

    const float* seg0 = a.seg[0].base + (long)nn * a.seg[0].stride_n + a.seg[0].set_off[set];
    const float* seg1 = a.seg[1].C > 0 ? a.seg[1].base + (long)nn * a.seg[1].stride_n + a.seg[1].set_off[set] : nullptr;
    const int C0 = a.seg[0].C;
    const float* aux = a.aux ? a.aux + (long)set * a.aux_set_stride : nullptr;
    const float* wsrc = a.w + (long)set * a.w_set_stride + (long)cb * nchunk * (9 * CK * BN);

    f32x16 acc[NCG][kQG];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int j = 0; j < kQG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.0f;

    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (t / 3) * Wp + (t % 3);

    // staging geometry: wave w stages input planes lc = w, w+4, ... of the chunk; a lane moves up to NJ float4
    // of a plane row.  All global loads of a stage are issued before the first LDS write so that a stage costs
    // ONE memory round trip (a rolled scalar copy loop costs one per element and was 60 % of this kernel).
    constexpr int RCI = (CK + kWaves - 1) / kWaves;
    constexpr int NJ = 4;                          // 4 * 64 float4 = 1024 floats >= BQ + 2*Wp + 2 up to Wp = 255
    constexpr int NWV = (9 * CK * BN / 4 + kThreads - 1) / kThreads;
    const int TL4 = TLp >> 2;
    const bool vec_ok = ((plane & 3) == 0) && (Wp <= 255);

    for (int c = 0; c < nchunk; ++c) {
        if (TRACE && tr && c < 8) tr[8 + 4 * c] = __builtin_amdgcn_s_memtime();
        if (vec_ok) {
            float4 iv[RCI * NJ + NWV];
#pragma unroll
            for (int r = 0; r < RCI; ++r) {
                const int lc = wave + r * kWaves;
                const int ci = c * CK + lc;
                const float* src = nullptr;
                if (lc < CK) {
                    if (ci < C0) src = seg0 + (long)ci * plane;
                    else if (ci < a.Cin) src = seg1 + (long)(ci - C0) * plane;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int i4 = lane + 64 * j;
                    const int q = q0 + 4 * i4;
                    const bool ok = (src != nullptr) && (i4 < TL4) && (q < plane);
                    iv[r * NJ + j] = *reinterpret_cast<const float4*>(ok ? src + q : seg0);
                }
            }
            const float4* ws = reinterpret_cast<const float4*>(wsrc + (long)c * (9 * CK * BN));
#pragma unroll
            for (int k = 0; k < NWV; ++k) {
                const int i = tid + k * kThreads;
                iv[RCI * NJ + k] = ws[i < 9 * CK * BN / 4 ? i : 0];
            }
            __syncthreads();                       // previous chunk's MFMA reads are done
            if (TRACE && tr && c < 8) tr[8 + 4 * c + 1] = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int r = 0; r < RCI; ++r) {
                const int lc = wave + r * kWaves;
                if (lc < CK) {
                    const int ci = c * CK + lc;
                    const bool has = ci < a.Cin;
                    float4* dst = reinterpret_cast<float4*>(in_tile + lc * TLp);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int i4 = lane + 64 * j;
                        const bool ok = has && (q0 + 4 * i4 < plane);
                        if (i4 < TL4) dst[i4] = ok ? iv[r * NJ + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
            float4* wd = reinterpret_cast<float4*>(w_tile);
#pragma unroll
            for (int k = 0; k < NWV; ++k) {
                const int i = tid + k * kThreads;
                if (i < 9 * CK * BN / 4) wd[i] = iv[RCI * NJ + k];
            }
        } else {
            __syncthreads();
            for (int lc = 0; lc < CK; ++lc) {      // generic (unaligned plane) fallback
                const int ci = c * CK + lc;
                const float* src = nullptr;
                if (ci < C0) src = seg0 + (long)ci * plane;
                else if (ci < a.Cin) src = seg1 + (long)(ci - C0) * plane;
                float* dst = in_tile + lc * TLp;
                for (int i = tid; i < TL; i += kThreads) {
                    const int q = q0 + i;
                    dst[i] = (src && q < plane) ? src[q] : 0.0f;
                }
            }
            const float4* ws = reinterpret_cast<const float4*>(wsrc + (long)c * (9 * CK * BN));
            float4* wd = reinterpret_cast<float4*>(w_tile);
            for (int i = tid; i < 9 * CK * BN / 4; i += kThreads) wd[i] = ws[i];
        }
        __syncthreads();
        if (TRACE && tr && c < 8) tr[8 + 4 * c + 2] = __builtin_amdgcn_s_memtime();
        // ---- MFMA ----
        const float* ibase = in_tile + hi * TLp + wave * (kQG * 32) + lo;
        const float* wbase = w_tile + hi * BN + lo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kp = 0; kp < CK / 2; ++kp) {
                float av[NCG], bv[kQG];
#pragma unroll
                for (int g = 0; g < NCG; ++g) av[g] = wbase[(t * CK + 2 * kp) * BN + g * 32];
#pragma unroll
                for (int j = 0; j < kQG; ++j) bv[j] = ibase[(2 * kp) * TLp + toff[t] + j * 32];
#pragma unroll
                for (int g = 0; g < NCG; ++g)
#pragma unroll
                    for (int j = 0; j < kQG; ++j)
                        acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g], bv[j], acc[g][j], 0, 0, 0);
            }
        }
        if (TRACE && tr && c < 8) tr[8 + 4 * c + 3] = __builtin_amdgcn_s_memtime();
    }
    if (TRACE && tr) tr[2] = __builtin_amdgcn_s_memtime();

    if constexpr (EPI <= EPI_SWISH) {
        conv_epilogue_flat<NCG, EPI>(a, acc, n, cb, bq, nblk_q, aux, tid, smem, TRACE && tr ? tr + 52 : nullptr);
        if (TRACE && tr) {
            tr[3] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tr[4] = __builtin_amdgcn_s_memtime();
            tr[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[6] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        }
        return;
    }
    conv_epilogue<NCG, EPI>(a, acc, n, cb, bq, nblk_q, aux, tid);
    if (TRACE && tr) {
        tr[3] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr[4] = __builtin_amdgcn_s_memtime();
    }
}

template <int CK, int NCG, int EPI>
hipError_t launch_t(const ConvArgs& a, const PackedConv& pw, int n, hipStream_t s) {
    constexpr int BN = NCG * 32;
    const int TL = kBQ + 2 * a.Wp + 2;
    const int TLp = (TL + 3) & ~3;
    // the LDS transpose of the flat epilogue (66 KB) exists in the GroupNorm layers only.  (Requested for every layer it held DSen2's
    // 32 -> 32 kernels -- 33 KB of staging, 151 VGPRs -- at two workgroups per CU instead of three; measured: 5.06 ms per tile either
    // way.  The traced launch shows why: 3 x 36.9 k of 133 k cycles per tile are MFMA issue, 83 % of the pipe, at a 1.8 GHz clock.)
    const size_t lds_main = (size_t)(CK * TLp + 9 * CK * BN) * sizeof(float);
    const size_t lds = EPI <= EPI_SWISH ? std::max(lds_main, kFlatLdsBytes) : lds_main;
    static LdsConfig lds_cfg;
    if (hipError_t e = lds_cfg.ensure(&conv3x3_f32<CK, NCG, EPI>, lds); e != hipSuccess) return e;
    const int nblk_q = conv_q_blocks(a.Hp, a.Wp);
    dim3 grid(nblk_q * pw.ncb * n);
    if constexpr ((NCG == 2 && ((CK == 10 && EPI == EPI_RAW) || (CK == 8 && EPI == EPI_SWISH))) || (NCG == 1 && CK == 8 && EPI == EPI_BIAS_RELU)) {   // probe aid: one traced launch
        static const char* trace_path = getenv("TTC_F32_TRACE");           // of the gates conv (or, TTC_F32_TRACE_EPI=2, of the first
        static const int trace_epi = [] { const char* e = getenv("TTC_F32_TRACE_EPI"); return e ? atoi(e) : (int)EPI_RAW; }();   // big U-Net block)
        static int trace_left = trace_path ? 1 : 0;
        if (trace_left > 0 && grid.x > 2000 && EPI == trace_epi) {
            trace_left--;
            static LdsConfig lds_tr;
            (void)lds_tr.ensure(&conv3x3_f32<CK, NCG, EPI, true>, lds);
            unsigned long long* d = nullptr;
            const size_t bytes = (size_t)grid.x * 64 * sizeof(unsigned long long);
            (void)hipStreamSynchronize(s);
            if (hipMalloc(&d, bytes) == hipSuccess) {
                (void)hipMemset(d, 0, bytes);
                ConvArgs b = a; b.trace = d;
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0, s);
                    hipLaunchKernelGGL((conv3x3_f32<CK, NCG, EPI, true>), grid, dim3(kThreads), lds, s, b, pw.nchunk, nblk_q, pw.ncb);
                    (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s); (void)hipEventElapsedTime(&ms, e0, e1);
                }
                std::vector<unsigned long long> h((size_t)grid.x * 64);
                (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost); (void)hipFree(d);
                if (FILE* f = fopen(trace_path, "wb")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
                fprintf(stderr, "[f32] traced launch (CK %d, epilogue %d, %d chunks): %.3f ms, grid %u -> %s\n", CK, EPI, pw.nchunk, ms, grid.x, trace_path);
            }
        }
    }
    hipLaunchKernelGGL((conv3x3_f32<CK, NCG, EPI>), grid, dim3(kThreads), lds, s, a, pw.nchunk, nblk_q, pw.ncb);
    return hipGetLastError();
}


// Narrow-cout form for DSen2's out_conv (32 -> 6, + bilinear + tanh): v_mfma_f32_16x16x4_f32 tiles (16 couts x 16 pixels x 4
// channels per instruction, 8 passes) instead of 32x32x2 -- with 6 real output channels a 32-row tile spends 81 % of its
// MFMAs on padding rows, a 16-row tile 62 %, at the same matrix-pipe rate: half the MFMA time.  Same staging and LDS image
// as conv3x3_f32 (the packed weights keep their 32-cout rows; rows 0..15 are read).  Operands: lane l -> row / column l & 15,
// k = l >> 4 (four channels of one tap per instruction); a wave owns 8 pixel blocks of 16 = its 128 positions.
// Accumulator: lane l holds couts 4 (l >> 4) + i, i = 0..3, of pixel l & 15.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int CK>
__global__ __launch_bounds__(kThreads, 3) void conv3x3_f32_head(ConvArgs a, int nchunk, int nblk_q, int ncb) {
    static_assert(CK % 4 == 0, "four channels per MFMA");
    constexpr int BN = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Wp = a.Wp, Hp = a.Hp;
    const int plane = Hp * Wp;
    const int TL = kBQ + 2 * Wp + 2;
    const int TLp = (TL + 3) & ~3;
    float* in_tile = smem;                   // [CK][TLp]
    float* w_tile = smem + CK * TLp;         // [9][CK][BN]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, ln = lane & 15, lk = lane >> 4;
    int bq, cb, n;
    tile_index(nblk_q, ncb, bq, cb, n);
    const int q0 = bq * kBQ;
    const float* seg0 = a.seg[0].base + (long)n * a.seg[0].stride_n + a.seg[0].set_off[0];
    const float* wsrc = a.w + (long)cb * nchunk * (9 * CK * BN);

    f32x4m acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = f32x4m{0.f, 0.f, 0.f, 0.f};
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (t / 3) * Wp + (t % 3);

    constexpr int RCI = (CK + kWaves - 1) / kWaves;
    constexpr int NJ = 4;
    constexpr int NWV = (9 * CK * BN / 4 + kThreads - 1) / kThreads;
    const int TL4 = TLp >> 2;
    for (int c = 0; c < nchunk; ++c) {
        float4 iv[RCI * NJ + NWV];                 // all global loads of the stage in flight before the first LDS write
#pragma unroll
        for (int r = 0; r < RCI; ++r) {
            const int lc = wave + r * kWaves;
            const int ci = c * CK + lc;
            const float* src = (lc < CK && ci < a.Cin) ? seg0 + (long)ci * plane : nullptr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i4 = lane + 64 * j;
                const int q = q0 + 4 * i4;
                const bool ok = (src != nullptr) && (i4 < TL4) && (q < plane);
                iv[r * NJ + j] = *reinterpret_cast<const float4*>(ok ? src + q : seg0);
            }
        }
        const float4* ws = reinterpret_cast<const float4*>(wsrc + (long)c * (9 * CK * BN));
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const int i = tid + k * kThreads;
            iv[RCI * NJ + k] = ws[i < 9 * CK * BN / 4 ? i : 0];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RCI; ++r) {
            const int lc = wave + r * kWaves;
            if (lc < CK) {
                const bool has = c * CK + lc < a.Cin;
                float4* dst = reinterpret_cast<float4*>(in_tile + lc * TLp);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int i4 = lane + 64 * j;
                    const bool ok = has && (q0 + 4 * i4 < plane);
                    if (i4 < TL4) dst[i4] = ok ? iv[r * NJ + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        float4* wd = reinterpret_cast<float4*>(w_tile);
#pragma unroll
        for (int k = 0; k < NWV; ++k) {
            const int i = tid + k * kThreads;
            if (i < 9 * CK * BN / 4) wd[i] = iv[RCI * NJ + k];
        }
        __syncthreads();
        const float* ibase = in_tile + lk * TLp + wave * (kQG * 32) + ln;
        const float* wbase = w_tile + lk * BN + ln;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < CK / 4; ++ks) {
                const float av = wbase[(t * CK + 4 * ks) * BN];
                float bv[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) bv[b] = ibase[(4 * ks) * TLp + toff[t] + 16 * b];
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[b], acc[b], 0, 0, 0);
            }
    }
    // epilogue: out = residual + tanh(conv + bias) for the couts this lane holds (lanes with l >> 4 >= 2 hold padding rows)
    const int Hout = Hp - 2, Wout = Wp - 2;
    float bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int co = 4 * lk + i; const float bvv = a.aux[co < a.Cout ? co : 0]; bias[i] = co < a.Cout ? bvv : 0.f; }
    float* outn = a.out + (long)n * a.out_stride_n;
    const float* resn = a.res + (long)n * a.out_stride_n;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int q = q0 + wave * (kQG * 32) + 16 * b + ln;
        const int y = q / Wp, x = q - y * Wp;
        const bool valid = (x < Wout) && (y < Hout);
        const long opix = (long)(y + a.oy) * a.out_pitch + (x + a.ox);
        float rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const bool ok = valid && (4 * lk + i < a.Cout); rv[i] = resn[ok ? (long)(4 * lk + i) * a.out_plane + opix : 0]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (valid && 4 * lk + i < a.Cout) outn[(long)(4 * lk + i) * a.out_plane + opix] = rv[i] + tanhf(acc[b][i] + bias[i]);
    }
}


// ---- DSen2's head 32 -> 6 (+ bias, tanh, + bilinear bands) on the VECTOR ALU (round 6) -----------------------------------------------------
// Six output channels fill 6 of the 16 rows of the narrowest fp32 MFMA tile: conv3x3_f32_head issues 2.7 x the algorithmic matrix work and ran
// 286 us per launch (0.57 ms of a tile's 5.05 ms of DSen2 convs).  The same sums cost 2 * 9 * 32 * 6 = 3456 flops per pixel on the vector ALU,
// whose packed fp32 FMA has the fp32 MFMA's peak rate: thread = 4 pixels of a row x 6 couts = 12 packed accumulators (cout pairs), weights are
// workgroup-uniform -> scalar loads straight from the layer's packed image (no LDS for them), inputs staged 4 channels at a time as whole
// padded rows (double-buffered, one barrier per chunk), each (channel, tap row) = one ds_read_b128 + one ds_read_b64 for 36 packed FMAs.
// Same products as the MFMA form, summed channel-major instead of tap-major: <= 1e-6 apart (tests/test_gpu_tile.py: DSen2 vs the oracle, 2e-5).
constexpr int kHvCk = 4, kHvRows = 8, kHvPitch = 136, kHvThreads = 256;
typedef float v2fh __attribute__((ext_vector_type(2)));
// tanh on v_exp_f32 / v_rcp_f32: 1 - 2 / (e^(2|x|) + 1), sign restored.  Absolute error <= ~1.2e-7 (one ulp of 1 from the final subtraction plus
// the 2-ulp relative error of v_exp_f32 scaled by 2 e^(2|x|) / (e^(2|x|) + 1)^2 <= 1/2), i.e. a few ulps of the reflectance it is added to; libm's
// tanhf is ~60 instructions and was a third of this kernel's time for 24 calls per thread
__device__ __forceinline__ float tanh_fast(float v) {
    const float e = __expf(2.0f * fabsf(v));
    return copysignf(1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f), v);
}
__global__ __launch_bounds__(kHvThreads, 2) void conv3x3_head_valu(ConvArgs a, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float st[2][kHvCk][kHvRows + 2][kHvPitch];
    __shared__ __attribute__((aligned(16))) float wl[32 * 9 * 8];             // the layer's weights [channel][tap][8: couts 0..5, 2 unused]
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int n = blockIdx.x / tiles_y, Y0 = (blockIdx.x - n * tiles_y) * kHvRows;
    const int Wp = a.Wp, Hp = a.Hp, W4 = Wp >> 2;
    const long plane = (long)Hp * Wp;
    const float* src = a.seg[0].base + (long)n * a.seg[0].stride_n + a.seg[0].set_off[0];
    const int nchunk = a.Cin / kHvCk;
    // columns Wp .. pitch-1 of every staged row are never written by the loads: they only feed pixels x >= W, which are not stored
    {
        const int ncol = kHvPitch - Wp;
        for (int i = tid; i < 2 * kHvCk * (kHvRows + 2) * ncol; i += kHvThreads) {
            const int r = i / ncol, cidx = Wp + i % ncol;
            (&st[0][0][0][0])[r * kHvPitch + cidx] = 0.f;
        }
    }
    // weights: packed image [chunk of 8][tap][8 channels][32 couts] -> LDS [channel][tap][8]; every lane of a wave reads the same address later
    for (int i = tid; i < a.Cin * 9 * 2; i += kHvThreads) {
        const int h = i & 1, ct = i >> 1, g = ct / 9, t = ct - g * 9;
        *reinterpret_cast<float4*>(&wl[ct * 8 + 4 * h]) =
            *reinterpret_cast<const float4*>(a.w + (((long)(g >> 3) * 9 + t) * 8 + (g & 7)) * 32 + 4 * h);
    }
    // staging pieces of this thread: the same (channel, row, float4 column) in every chunk
    constexpr int NLD = 5;                                        // 4 channels x 10 rows x Wp / 4 <= 5 x 256 pieces: Wp <= 128
    const int per_chunk = kHvCk * (kHvRows + 2) * W4;
    int goff[NLD], loff[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * kHvThreads;
        const int ch = i / ((kHvRows + 2) * W4), rem = i - ch * ((kHvRows + 2) * W4), r = rem / W4, c4 = rem - r * W4;
        const bool ok = i < per_chunk && (Y0 + r) < Hp;
        goff[k] = ok ? (int)(ch * plane + (long)(Y0 + r) * Wp + 4 * c4) : -1;
        loff[k] = i < per_chunk ? (ch * (kHvRows + 2) + r) * kHvPitch + 4 * c4 : -1;
    }
    float4 pre[2][NLD];                                           // TWO chunks in flight: a chunk's loads get a whole chunk of arithmetic to land
    auto fetch = [&](int c, int slot) {
        const float* base = src + (long)c * kHvCk * plane;
#pragma unroll
        for (int k = 0; k < NLD; ++k) pre[slot][k] = goff[k] >= 0 ? *reinterpret_cast<const float4*>(base + goff[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stash = [&](int buf, int slot) {
        float* dst = &st[buf][0][0][0];
#pragma unroll
        for (int k = 0; k < NLD; ++k) if (loff[k] >= 0) *reinterpret_cast<float4*>(dst + loff[k]) = pre[slot][k];
    };
    v2fh acc[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[j][p] = v2fh{0.f, 0.f};
    fetch(0, 0);
    if (nchunk > 1) fetch(1, 1);
    stash(0, 0);
    __syncthreads();
    for (int c0 = 0; c0 < nchunk; c0 += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {                       // chunk parity is static: register sets and LDS buffers need no moves
            const int c = c0 + par;
            if (c < nchunk) {
                if (c + 2 < nchunk) fetch(c + 2, par);            // set `par` was stored to LDS one chunk ago
#pragma unroll 1
                for (int ch = 0; ch < kHvCk; ++ch) {              // not unrolled: unrolled, the scheduler hoists every tap's weight reads (> 256 VGPRs, spills)
                    const float* wg = wl + (c * kHvCk + ch) * 72;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const float* row = &st[par][0][ty + dy][4 * tx] + ch * ((kHvRows + 2) * kHvPitch);
                        const float4 xa = *reinterpret_cast<const float4*>(row);
                        const float2 xb = *reinterpret_cast<const float2*>(row + 4);
                        const float x[6] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y};
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float4 wa = *reinterpret_cast<const float4*>(wg + (dy * 3 + dx) * 8);          // wave-uniform address: one broadcast read
                            const float2 wb = *reinterpret_cast<const float2*>(wg + (dy * 3 + dx) * 8 + 4);
                            const v2fh w01 = v2fh{wa.x, wa.y}, w23 = v2fh{wa.z, wa.w}, w45 = v2fh{wb.x, wb.y};
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                const v2fh xx = v2fh{x[dx + p], x[dx + p]};
                                acc[0][p] = __builtin_elementwise_fma(w01, xx, acc[0][p]);
                                acc[1][p] = __builtin_elementwise_fma(w23, xx, acc[1][p]);
                                acc[2][p] = __builtin_elementwise_fma(w45, xx, acc[2][p]);
                            }
                        }
                    }
                }
                if (c + 1 < nchunk) stash(par ^ 1, par ^ 1);      // chunk c + 1, requested a whole chunk ago
                __syncthreads();
            }
        }
    }
    const int Hout = Hp - 2, Wout = Wp - 2, y = Y0 + ty;
    if (y >= Hout) return;
    float* outn = a.out + (long)n * a.out_stride_n;
    const float* resn = a.res + (long)n * a.out_stride_n;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = 4 * tx + p;
        if (x >= Wout) continue;
        const long opix = (long)(y + a.oy) * a.out_pitch + (x + a.ox);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long o0 = (long)(2 * j) * a.out_plane + opix, o1 = o0 + a.out_plane;
            outn[o0] = resn[o0] + tanh_fast(acc[j][p].x + a.aux[2 * j]);
            outn[o1] = resn[o1] + tanh_fast(acc[j][p].y + a.aux[2 * j + 1]);
        }
    }
}
// the launch limits of the vector-ALU head; TTC_HEAD_VALU=0 keeps the MFMA form (A/B runs)
static bool head_valu_ok(const ConvArgs& a, const PackedConv& pw, int epi, int n) {
    static const int on = [] { const char* e = getenv("TTC_HEAD_VALU"); return e ? atoi(e) : 1; }();
    return on && pw.mode == 0 && epi == EPI_BIAS_TANH_ADD && a.Cout == 6 && pw.CK == 8 && pw.BN == 32 && pw.ncb == 1 && a.seg[1].C == 0 &&
           a.Cin == 32 && (a.Wp & 3) == 0 && a.Wp <= 128 && (long)a.Cin * a.Hp * a.Wp < (1L << 31) && a.n_per_set >= n && !a.reflect_out && a.res != nullptr &&
           (long)n * ((a.Hp - 2 + kHvRows - 1) / kHvRows) < (1L << 31);
}
template <int CK>
hipError_t launch_head(const ConvArgs& a, const PackedConv& pw, int n, hipStream_t s) {
    const int TL = kBQ + 2 * a.Wp + 2;
    const int TLp = (TL + 3) & ~3;
    const size_t lds = (size_t)(CK * TLp + 9 * CK * 32) * sizeof(float);
    static LdsConfig lds_cfg;
    if (hipError_t e = lds_cfg.ensure(&conv3x3_f32_head<CK>, lds); e != hipSuccess) return e;
    const int nblk_q = conv_q_blocks(a.Hp, a.Wp);
    hipLaunchKernelGGL((conv3x3_f32_head<CK>), dim3(nblk_q * pw.ncb * n), dim3(kThreads), lds, s, a, pw.nchunk, nblk_q, pw.ncb);
    return hipGetLastError();
}

}  // namespace

int conv_q_blocks(int Hp, int Wp) { return ((Hp - 2) * Wp + kBQ - 1) / kBQ; }
int conv_stat_slots(int Hp, int Wp) { return conv_q_blocks(Hp, Wp) * kWaves; }

// TTC_WINOGRAD=0 keeps every fp32 layer on the direct kernel (A/B runs, parity tests of the direct form); TTC_WINO4=0 keeps the
// F(2x2, 3x3) kernels for the layers the F(4x4, 3x3) form would take
bool conv_use_wino(const PackedConv& pw, int epi) {
    static const int on = [] { const char* e = getenv("TTC_WINOGRAD"); return e ? atoi(e) : 1; }();
    if (!on || pw.mode != 0 || pw.form >= 2 || pw.d_wu == nullptr || pw.nchunk_w < 3) return false;
    // GroupNorm layers of the ConvGRU / U-Net only.  DSen2's 32 -> 32 layers were tried on the same kernel (bias / ReLU / residual
    // epilogues with the reflect rim) and measured SLOWER than the direct form (2.6 / 3.1 vs 1.9 / 2.2 ms per tile: 118-px windows
    // are 7.4 regions wide (15 % padding), a tile is only four chunks long, and the padded-plane output forces scalar stores):
    // profiles/r04_dsen2_winograd_probe_kernel_stats.md, csrc/experiments/README.md
    return epi <= EPI_SWISH && pw.Cout % 32 == 0;
}
// the launch limits of the F(2x2) kernels (conv3x3_wino.hip launch_w), as a predicate: a configuration outside them falls back to the
// direct kernel instead of failing the forward with hipErrorInvalidValue
static bool conv_wino2_ok(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set) {
    if (!conv_use_wino(pw, epi) || (Wp & 1)) return false;
    if ((long)Hp * Wp >= (1L << 24) || (long)Cin * Hp * Wp >= (1L << 31) || n_per_set < 1 || n_per_set >= 4096) return false;
    const int ncb = pw.Cout >= 64 ? 2 : 1, tb = 2 / ncb;
    const long TX = (Wp - 2 + 1) / 2, TY = (Hp - 2 + 1) / 2;
    const long RXn = (TX + 7) / 8, RYn = (TY + 4 * tb - 1) / (4 * tb), ncp = (pw.Cout + 32 * ncb - 1) / (32 * ncb);
    return RXn < 4096 && RYn < 4096 && ncp < 4096 && RXn * RYn * ncp * n < (1L << 24);
}
ConvKernel conv_kernel_for(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set) {
    static const int on4 = [] { const char* e = getenv("TTC_WINO4"); return e ? atoi(e) : 1; }();
    static const int on = [] { const char* e = getenv("TTC_WINOGRAD"); return e ? atoi(e) : 1; }();
    if (on && on4 && pw.mode == 0 && pw.form == 0 && conv_wino4_ok(pw, epi, Hp, Wp, Cin, n, n_per_set)) return CONV_WINO4;
    if (conv_wino2_ok(pw, epi, Hp, Wp, Cin, n, n_per_set)) return CONV_WINO2;
    return CONV_DIRECT;
}
int conv_stat_slots_for(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set) {
    switch (conv_kernel_for(pw, epi, Hp, Wp, Cin, n, n_per_set)) {
        case CONV_WINO4: return conv_wino4_stat_slots(Hp, Wp);
        case CONV_WINO2: return conv_wino_stat_slots(Hp, Wp, pw.Cout);
        default: return conv_stat_slots(Hp, Wp);
    }
}

int conv_pick_ck(int Cin) {
    // smallest K padding: 49 -> 5 x 10, 17 -> 3 x 6, 10 -> 1 x 10, multiples of 8 -> 8
    if (Cin % 8 == 0) return 8;
    const int p8 = ((Cin + 7) / 8) * 8, p10 = ((Cin + 9) / 10) * 10, p6 = ((Cin + 5) / 6) * 6;
    if (p6 <= p8 && p6 <= p10) return 6;
    return p10 <= p8 ? 10 : 8;
}

int conv_pick_bn(int Cout) { return Cout >= 64 ? 64 : 32; }

long conv_pack(const float* const* hwio, int nsets, int Cin, int Cout, int CK, int BN, std::vector<float>& out) {
    const int nchunk = (Cin + CK - 1) / CK, ncb = (Cout + BN - 1) / BN;
    const long per_set = (long)ncb * nchunk * 9 * CK * BN;
    out.assign((size_t)per_set * nsets, 0.0f);
    for (int s = 0; s < nsets; ++s)
        for (int cb = 0; cb < ncb; ++cb)
            for (int c = 0; c < nchunk; ++c)
                for (int t = 0; t < 9; ++t)
                    for (int lc = 0; lc < CK; ++lc)
                        for (int co = 0; co < BN; ++co) {
                            const int ci = c * CK + lc, o = cb * BN + co;
                            if (ci >= Cin || o >= Cout) continue;
                            // HWIO: [(t/3)][(t%3)][ci][o]
                            out[(size_t)s * per_set + ((((long)cb * nchunk + c) * 9 + t) * CK + lc) * BN + co] =
                                hwio[s][(((t / 3) * 3 + (t % 3)) * Cin + ci) * (long)Cout + o];
                        }
    return per_set;
}

ttc_status conv_upload(ttc_ctx* c, PackedConv& pc, const float* const* hwio, int nsets, int Cin, int Cout, int BN, int C0, int mode) {
    pc.Cin = Cin; pc.Cout = Cout; pc.nsets = nsets;
    pc.CK = conv_pick_ck(Cin); pc.BN = BN;
    pc.nchunk = (Cin + pc.CK - 1) / pc.CK; pc.ncb = (Cout + pc.BN - 1) / pc.BN;
    pc.mode = mode < 0 ? c->cfg.precision : mode;
    pc.form = c->cfg.fp32_conv_form;
    std::vector<float> packed;
    pc.set_stride = conv_pack(hwio, nsets, Cin, Cout, pc.CK, pc.BN, packed);
    if (!pc.d_w && !(pc.d_w = c->alloc_f(packed.size()))) return c->fail(TTC_ERR_NOMEM, "hipMalloc weights");
    TTC_HIP(c, hipMemcpy(pc.d_w, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    if (pc.mode == 0 && Cout % 32 == 0) {   // fp32 engine: Winograd images for the GroupNorm layers (conv3x3_wino.hip)
        std::vector<float> pu;
        pc.set_stride_w = conv_pack_wino(hwio, nsets, Cin, Cout, pu, &pc.nchunk_w);
        if (!pc.d_wu && !(pc.d_wu = c->alloc_f(pu.size()))) return c->fail(TTC_ERR_NOMEM, "hipMalloc Winograd weights");
        TTC_HIP(c, hipMemcpy(pc.d_wu, pu.data(), pu.size() * sizeof(float), hipMemcpyHostToDevice));
        if (Cout % 64 == 0) {           // ... and the F(4x4, 3x3) images for the 64-cout-multiple layers (conv3x3_wino4.hip)
            pc.set_stride_w4 = conv_pack_wino4(hwio, nsets, Cin, Cout, pu, &pc.nchunk_w4);
            if (!pc.d_wu4 && !(pc.d_wu4 = c->alloc_f(pu.size()))) return c->fail(TTC_ERR_NOMEM, "hipMalloc Winograd F(4x4) weights");
            TTC_HIP(c, hipMemcpy(pc.d_wu4, pu.data(), pu.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (pc.mode >= 2) {                 // 16-bit engine: fp16 (2) / bf16 (3) hi | lo LDS images
        std::vector<uint16_t> ph;
        pc.set_stride_h = conv_pack_h16(hwio, nsets, Cin, C0 < 0 ? Cin : C0, Cout, pc.BN, pc.mode == 3, ph, &pc.nchunk_h);
        // + one DMA piece of slack: the last 1-KiB piece of a 32-cout plane is only half used
        if (!pc.d_wh && !(pc.d_wh = reinterpret_cast<uint4*>(c->alloc_f((ph.size() + 1) / 2 + 256))))
            return c->fail(TTC_ERR_NOMEM, "hipMalloc 16-bit weights");
        TTC_HIP(c, hipMemcpy(pc.d_wh, ph.data(), ph.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    return TTC_OK;
}

double conv_issued_flops(const ConvArgs& a, const PackedConv& pw, int epi, int n) {
    const int H = a.Hp - 2, W = a.Wp - 2;
    const int cin_run = (a.cin_run > 0 && a.cin_run < a.Cin) ? a.cin_run : a.Cin;
    switch (conv_kernel_for(pw, epi, a.Hp, a.Wp, a.Cin, n, a.n_per_set)) {
        case CONV_WINO4: {       // conv3x3_wino4.hip launch_w4: tile = two 16 x 16-pixel sub-regions x 64 couts, 8 waves, v_mfma_f32_16x16x4_f32
            const long RR = (long)((W + 15) / 16) * ((H + 15) / 16), nsets = (n + a.n_per_set - 1) / a.n_per_set;
            const long pps = (RR * std::min(n, a.n_per_set) + 1) / 2, ntiles = pps * nsets * (a.Cout / 64);
            const int nrun = std::max(3, (cin_run + 7) / 8), rem = std::min(a.Cin, nrun * 8) - 8 * (nrun - 1), nks_last = (rem + 3) / 4;
            return (double)ntiles * 8.0 * ((nrun - 1) * 72.0 + nks_last * 36.0) * (2.0 * 16 * 16 * 4);
        }
        case CONV_WINO2: {       // conv3x3_wino.hip launch_w: tile = 32 * TB Winograd tiles x 32 * NCB couts, 4 waves, v_mfma_f32_32x32x2_f32
            const int NCB = pw.Cout >= 64 ? 2 : 1, TB = 2 / NCB, TX = (W + 1) / 2, TY = (H + 1) / 2;
            const long ntiles = (long)((TX + 7) / 8) * ((TY + 4 * TB - 1) / (4 * TB)) * ((a.Cout + 32 * NCB - 1) / (32 * NCB)) * n;
            const int nrun = std::max(3, (cin_run + 7) / 8), rem = std::min(a.Cin, nrun * 8) - 8 * (nrun - 1), nks_last = (rem + 1) / 2;
            return (double)ntiles * 4.0 * ((nrun - 1) * 32.0 + nks_last * 8.0) * (2.0 * 32 * 32 * 2);
        }
        default: break;
    }
    if (head_valu_ok(a, pw, epi, n)) return 0.0;          // the 32 -> 6 head runs on the vector ALU: no matrix instruction is issued
    // direct implicit GEMM: tile = 512 flat positions x BN couts (the 32 -> 6 head: 16 rows), K = 9 taps x nchunk x CK channels
    const double tiles = (double)conv_q_blocks(a.Hp, a.Wp) * pw.ncb * n;
    const bool head = pw.CK == 8 && pw.BN == 32 && epi == EPI_BIAS_TANH_ADD && pw.Cout <= 16 && pw.ncb == 1 && a.seg[1].C == 0 &&
                      ((a.Hp * a.Wp) & 3) == 0 && a.Wp <= 255 && a.n_per_set >= n;
    return tiles * 2.0 * (9.0 * pw.nchunk * pw.CK) * kBQ * (head ? 16 : pw.BN);
}

hipError_t conv_launch(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s) {
    switch (conv_kernel_for(pw, epi, a.Hp, a.Wp, a.Cin, n, a.n_per_set)) {
        case CONV_WINO4: return conv_launch_wino4(a, pw, epi, n, s);
        case CONV_WINO2: return conv_launch_wino(a, pw, epi, n, s);
        default: break;
    }
    // only the (CK, BN, epilogue) combinations the two graphs need are instantiated
#define TTC_CONV_CASE(ck, ncg, e) \
    if (pw.CK == ck && pw.BN == ncg * 32 && epi == e) return launch_t<ck, ncg, e>(a, pw, n, s);
    TTC_CONV_CASE(10, 2, EPI_RAW)            // ConvGRU gates        49 -> 64
    TTC_CONV_CASE(10, 1, EPI_SSE)            // ConvGRU candidate    49 -> 32
    TTC_CONV_CASE(6, 2, EPI_SWISH)           // conv_median          17 -> 64
    TTC_CONV_CASE(8, 2, EPI_SWISH)           // U-Net blocks         {64,128,256} -> {64,128,256}
    TTC_CONV_CASE(10, 1, EPI_BIAS_RELU)      // DSen2 in_conv        10 -> 32
    TTC_CONV_CASE(8, 1, EPI_BIAS_RELU)       // DSen2 x1_conv        32 -> 32
    TTC_CONV_CASE(8, 1, EPI_BIAS_RES)        // DSen2 x2_conv        32 -> 32 (+ residual)
    if (head_valu_ok(a, pw, epi, n)) {      // DSen2 out_conv 32 -> 6 (+ bilinear) on the vector ALU
        const int tiles_y = (a.Hp - 2 + kHvRows - 1) / kHvRows;
        hipLaunchKernelGGL(conv3x3_head_valu, dim3((unsigned)(n * tiles_y)), dim3(kHvThreads), 0, s, a, tiles_y);
        return hipGetLastError();
    }
    {   // DSen2 out_conv 32 -> 6 (+ bilinear): the 16-row MFMA form when the plane is aligned and has one input segment
        static const int narrow = [] { const char* e = getenv("TTC_CONV_NARROW"); return e ? atoi(e) : 1; }();
        if (narrow && pw.CK == 8 && pw.BN == 32 && epi == EPI_BIAS_TANH_ADD && pw.Cout <= 16 && pw.ncb == 1 && a.seg[1].C == 0 &&
            ((a.Hp * a.Wp) & 3) == 0 && a.Wp <= 255 && a.n_per_set >= n)
            return launch_head<8>(a, pw, n, s);
    }
    TTC_CONV_CASE(8, 1, EPI_BIAS_TANH_ADD)   // DSen2 out_conv       32 -> 6  (+ bilinear)
#undef TTC_CONV_CASE
    return hipErrorInvalidValue;
}
