// Implicit-GEMM 3x3 convolution on the gfx950 16-bit matrix cores (v_mfma_f32_32x32x16_f16 / _bf16, fp32 accumulate),
// fed by CHANNEL-BLOCKED 16-bit activations that their producers already wrote in MFMA operand order.
// ttc_config.precision = 2 (fp16 elements) or 3 (bf16 elements); replaces the tf.nn.convolution / Conv2D nodes of the two
// reference graphs (src/train/src/model.py:251, :276, :416-442; superresolve_graph.pb) like the fp32 engine does.
//
// Operands.  x = x_hi + x_lo with two 16-bit elements (fp16 pair: 22 mantissa bits, bf16 pair: 16), weights likewise
// (split on the host).  A layer runs with TERMS = 3 products per K block,
//     x * w ~= x_lo*w_hi + x_hi*w_lo + x_hi*w_hi          (drops x_lo*w_lo ~ 2^-22 relative for fp16),
// or TERMS = 1 (x_hi*w_hi: plain 16-bit operands, 2^-11 relative for fp16) where the per-layer precision map allows it
// (tools/study/precision_study.py; on a real tile even the ConvGRU gates conv alone costs 3e-3: the default map is all-3).
//
// Layout.  Activations in HBM: [n][C8][Hp*Wp][8 ch] 16-bit, one 16-byte K vector per (channel block, position), hi and lo
// tensors.  Same "flattened padded plane" formulation as conv3x3_mfma.hip: a workgroup (4 waves) owns 512 consecutive
// positions q of one window x BN = 32*NCG output channels; a tap is a linear offset.  Per 8-channel chunk the tile
// [512 + 2*Wp + 2 positions][16 B] is ONE contiguous run of the blocked tensor, so staging is nothing but
// global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPRs, no VALU, no ds_write): the LDS image is lane-linear
// and consecutive positions are consecutive 16-byte slots, which ds_read_b128's 16-lane groups read conflict-free.
// Weights: host-packed LDS image [tap][cout][8 ch] per chunk and plane, copied the same way.
// One MFMA K block (16) = the 8 channels of TWO taps (half-wave 0: tap 2*kb, half-wave 1: tap 2*kb+1; the tenth tap
// half has a zero A operand): 5 K blocks per chunk.
//
// Schedule (2 workgroups per CU, <= 80 KB LDS each):
//   TERMS = 1: 3-stage LDS ring, prefetch distance 2, ONE barrier per chunk, counted vmcnt (the DMA of chunk c+1 stays
//              in flight across the barrier).
//   TERMS = 3: hi tiles double-buffered, lo tile single (it is read by one of the three products only: that product
//              runs first, a mid-chunk barrier frees the buffer and the next chunk's lo tile streams in under the other
//              two products), weights (hi | lo) double-buffered.
//
// Round 3: what separated the round-2 kernel from its matrix-pipe floor was not the chunk loop (two co-resident waves keep
// the pipe of a SIMD busy while both are inside it) but the 35 % of a workgroup's life spent OUTSIDE it -- kernel entry,
// index math, the first copies' round trip, and an epilogue that transposed the tile through LDS behind four barriers
// (per-workgroup s_memtime traces).  Two structural changes attack exactly that:
//   * PERSISTENT workgroups (grid = the resident set, 2 per CU) walk the XCD-aware tile list, and the chunk stream runs
//     ACROSS tiles: while the last chunk of tile i is being multiplied, the copies of chunk 0 of tile i + 1 are already in
//     flight into the free halves of the double buffers, so they land under tile i's epilogue instead of after a new
//     workgroup's start-up;
//   * NO LDS in any epilogue: the GroupNorm layers' raw outputs leave straight from the accumulators as 16-byte vectors in
//     the same channel-blocked layout as every other tensor of this engine -- EXACT fp32, stored as a plane of top and a
//     plane of bottom 16-bit halves (cross-half v_permlane32_swap assembles 8 consecutive channels per lane) -- which is
//     what lets the prefetch above own the LDS during the epilogue, and removes 256 ds_write_b32 + 32 ds_read_b128 +
//     4 barriers per tile.  The consumers (k_gru_apply*_b16, k_block_finalize_b16, k_head, k_tap_late) read those planes
//     as 16-byte vectors (raw_load8, h16_common.h).
#include <atomic>
#include <algorithm>

#include "h16_common.h"

using namespace ttcconv;

namespace {

constexpr int kKB = 5;                      // K blocks per 8-channel chunk (tap pairs)

// raw s_barrier (no vmcnt drain: LDS-DMA stays in flight across it) fenced against compiler reordering of LDS accesses
__device__ __forceinline__ void cbarrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Fused epilogue: EPI op, then the fp32 accumulators leave as channel-blocked 16-byte K vectors.  A lane holds 4 of a
// block's 8 channels (rows (r & 3) + 8 * (r >> 2) + 4 * hi): v_permlane32_swap exchanges halves between lane and lane + 32,
// after which the low half-wave owns channel block 2k and the high half-wave block 2k + 1 of 32 consecutive positions
// -> 512 contiguous bytes per half-wave and store.
//   EPI >= EPI_BIAS (DSen2): the NEXT layer's input, hi + lo 16-bit pair, in its padded plane (reflect rim duplicated);
//   EPI <= EPI_SWISH (GroupNorm layers): the RAW output, exact fp32 as top / bottom 16-bit halves (o_hi = top plane,
//   o_lo = bottom plane), at the tile's own flat positions q (the output keeps the input pitch; the two junk columns per
//   row are written and never read), plus the deterministic GroupNorm partial sums of conv_common.h.
template <int BF, int NCG, int EPI>
__device__ __forceinline__ void h16_epilogue_b16(const H16Args& a, f32x16 (&acc)[NCG][kQG], int n, int cb, int bq, int nblk_q,
                                                 const float* aux, int tid) {
    using E = Elem<BF>;
    constexpr int BN = NCG * 32;
    constexpr bool GN = EPI <= EPI_SWISH;
    const ConvArgs& c = a.c;
    const int Wp = c.Wp, Hp = c.Hp;
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int q0 = bq * kBQ;
    const int Hout = Hp - 2, Wout = Wp - 2;
    const int C8out = (c.Cout + 7) >> 3;
    const long qend = (long)Hout * Wp;            // GN layers: flat positions of the rows that exist
    float ssum[NCG][4], ssq[NCG][4];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) { ssum[g][k] = 0.f; ssq[g][k] = 0.f; }
    uint4* ohi = a.o_hi + (long)n * a.o_stride_n;
    uint4* olo = a.o_lo ? a.o_lo + (long)n * a.o_stride_n : nullptr;
    const uint2* rhi = (EPI == EPI_BIAS_RES) ? reinterpret_cast<const uint2*>(a.r_hi + (long)n * a.o_stride_n) : nullptr;
    const uint2* rlo = (EPI == EPI_BIAS_RES) ? reinterpret_cast<const uint2*>(a.r_lo + (long)n * a.o_stride_n) : nullptr;
    float bias[NCG][16];          // fetched once, unconditionally (see conv_epilogue)
    if (EPI >= EPI_BIAS) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float bv = aux[co < c.Cout ? co : 0];
                bias[g][r] = co < c.Cout ? bv : 0.f;
            }
    }
    float k1[16];                 // EPI_SSE: the in-cell sSE kernel of this lane's 16 channels
    if (EPI == EPI_SSE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) k1[r] = aux[(r & 3) + 8 * (r >> 2) + 4 * hi];
    }

#pragma unroll
    for (int j = 0; j < kQG; ++j) {
        const int q = q0 + (wave * kQG + j) * 32 + lo;
        const int y = q / Wp, x = q - y * Wp;
        const bool valid = (x < Wout) && (y < Hout);
        const long opix = GN ? (long)q : (long)(y + c.oy) * c.out_pitch + (x + c.ox);
        const bool okpix = GN ? ((long)q < qend) : valid;
        long dup_y = 0, dup_x = 0;
        bool any_dup = false;
        if (EPI >= EPI_BIAS && c.reflect_out) {
            if (valid) {
                dup_y = (y == 1) ? -2L * c.out_pitch : ((y == Hout - 2) ? 2L * c.out_pitch : 0L);
                dup_x = (x == 1) ? -2L : ((x == Wout - 2) ? 2L : 0L);
            }
            any_dup = __any((dup_y != 0) || (dup_x != 0));
        }
        float gate = 1.0f;
        if (EPI == EPI_SSE) {
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += k1[r] * acc[0][j][r];
            dot += __shfl_xor(dot, 32);
            gate = sigmoidf_(dot);
        }
        float ratio = 1.0f;
        if (EPI == EPI_SWISH && c.same_pad) {
            const bool ey = (y == 0) || (y == Hout - 1), ex = (x == 0) || (x == Wout - 1);
            ratio = (ey && ex) ? 2.25f : ((ey || ex) ? 1.5f : 1.0f);
        }
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            float v[16];
            if (EPI == EPI_BIAS_RES) {
                // residual x = hi + lo of this lane's own 4 channels per block: 8-byte halves of the blocked K vectors
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int blk = cb * (BN / 8) + g * 4 + k;
                    const bool ok = valid && (blk < C8out);
                    const long u = ok ? ((long)blk * a.o_plane + opix) * 2 + hi : 0L;
                    const uint2 h2 = rhi[u], l2 = rlo[u];
                    float t0, t1;
                    E::unpack2(h2.x, v[4 * k + 0], v[4 * k + 1]); E::unpack2(l2.x, t0, t1); v[4 * k + 0] += t0; v[4 * k + 1] += t1;
                    E::unpack2(h2.y, v[4 * k + 2], v[4 * k + 3]); E::unpack2(l2.y, t0, t1); v[4 * k + 2] += t0; v[4 * k + 3] += t1;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float t = acc[g][j][r];
                if (EPI == EPI_SSE) t *= gate;
                if (EPI == EPI_SWISH) { t *= ratio; t = t * sigmoidf_(t); }
                if (EPI >= EPI_BIAS) {
                    t += bias[g][r];
                    if (EPI == EPI_BIAS_RELU) t = fmaxf(t, 0.f);
                    if (EPI == EPI_BIAS_RES) t = v[r] + 0.1f * t;
                }
                if (co >= c.Cout) t = 0.f;                      // pad channels of the last block stay zero
                v[r] = t;
                if (GN && valid) { ssum[g][r >> 2] += t; ssq[g][r >> 2] += t * t; }
            }
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {                    // block pairs (2kp, 2kp + 1) of this cout group
                unsigned xl0, xl1, yl0, yl1, xh0, xh1, yh0, yh1;
                if (GN) {
                    xh0 = raw_top2(v[8 * kp + 0], v[8 * kp + 1]); xl0 = raw_bot2(v[8 * kp + 0], v[8 * kp + 1]);
                    xh1 = raw_top2(v[8 * kp + 2], v[8 * kp + 3]); xl1 = raw_bot2(v[8 * kp + 2], v[8 * kp + 3]);
                    yh0 = raw_top2(v[8 * kp + 4], v[8 * kp + 5]); yl0 = raw_bot2(v[8 * kp + 4], v[8 * kp + 5]);
                    yh1 = raw_top2(v[8 * kp + 6], v[8 * kp + 7]); yl1 = raw_bot2(v[8 * kp + 6], v[8 * kp + 7]);
                } else {
                    xh0 = E::pack2(v[8 * kp + 0], v[8 * kp + 1], xl0); xh1 = E::pack2(v[8 * kp + 2], v[8 * kp + 3], xl1);
                    yh0 = E::pack2(v[8 * kp + 4], v[8 * kp + 5], yl0); yh1 = E::pack2(v[8 * kp + 6], v[8 * kp + 7], yl1);
                }
                // lanes 32-63 of X <-> lanes 0-31 of Y: afterwards [X | Y] = the 8 channels of block 2kp + hi
                auto s0 = __builtin_amdgcn_permlane32_swap(xh0, yh0, false, false); xh0 = s0[0]; yh0 = s0[1];
                auto s1 = __builtin_amdgcn_permlane32_swap(xh1, yh1, false, false); xh1 = s1[0]; yh1 = s1[1];
                const int blk = cb * (BN / 8) + g * 4 + 2 * kp + hi;
                const bool ok = okpix && (blk < C8out);
                const long u = (long)blk * a.o_plane + opix;
                const uint4 vh = make_uint4(xh0, xh1, yh0, yh1);
                if (ok) {
                    ohi[u] = vh;
                    if (any_dup) {
                        if (dup_y) ohi[u + dup_y] = vh;
                        if (dup_x) ohi[u + dup_x] = vh;
                        if (dup_y && dup_x) ohi[u + dup_y + dup_x] = vh;
                    }
                }
                if (olo) {
                    auto t0 = __builtin_amdgcn_permlane32_swap(xl0, yl0, false, false); xl0 = t0[0]; yl0 = t0[1];
                    auto t1 = __builtin_amdgcn_permlane32_swap(xl1, yl1, false, false); xl1 = t1[0]; yl1 = t1[1];
                    const uint4 vl = make_uint4(xl0, xl1, yl0, yl1);
                    if (ok) {
                        olo[u] = vl;
                        if (any_dup) {
                            if (dup_y) olo[u + dup_y] = vl;
                            if (dup_x) olo[u + dup_x] = vl;
                            if (dup_y && dup_x) olo[u + dup_y + dup_x] = vl;
                        }
                    }
                }
            }
        }
    }

    if (GN && c.stats) {        // same deterministic GroupNorm partials as conv_common.h
        float red[NCG * 8];
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(g * 4 + k) * 2] = ssum[g][k]; red[(g * 4 + k) * 2 + 1] = ssq[g][k]; }
        half_wave_sums(red);
        if (lo == 31) {           // one exec region for the two writer lanes; (sum, sumsq) leave as one 8-byte store
            const long slots = (long)nblk_q * kWaves;
            float2* base = reinterpret_cast<float2*>(c.stats) + (long)n * (c.Cout / 4) * slots + bq * kWaves + wave;
#pragma unroll
            for (int g = 0; g < NCG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int quad = cb * (BN / 4) + g * 8 + 2 * k + hi;
                    if (quad * 4 < c.Cout) base[quad * slots] = make_float2(red[(g * 4 + k) * 2], red[(g * 4 + k) * 2 + 1]);
                }
        }
    }
}

template <int BF, int TERMS, int NCG, int EPI, int OUT, bool TRACE = false>
__global__ __launch_bounds__(kThreads, 2) void conv3x3_h16(H16Args a, int nblk_q, int ncb, int ntiles, int desync,
                                                            unsigned long long* __restrict__ trace) {
    using E = Elem<BF>;
    using v8 = typename E::v8;
    constexpr int BN = NCG * 32;
    constexpr int WPIECES = (9 * BN * 16 + 1023) / 1024;     // 1-KiB DMA pieces of one weight plane of a chunk
    constexpr int WUNITS = WPIECES * 64;                     // 16-byte units of that plane (padded)
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    // Persistent loop + ~100 dwords of arguments: if the loop body reads `a` directly, hipcc keeps every field it uses anywhere
    // live in SGPRs across the whole walk (measured: 124 SGPR + 166 VGPR spills).  The arguments are therefore re-read from the
    // kernarg segment (scalar loads, constant cache) at the start of each phase, through a pointer the optimiser cannot see
    // through: a phase's fields die with it.
    typedef const __attribute__((address_space(4))) H16Args* KArgs;
    const KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();                       // `a` is the first kernel argument
    auto args = [&]() { KArgs q = kp; asm volatile("" : "+s"(q)); return q; };
    auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };
    const int Wp = args()->c.Wp, Hp = args()->c.Hp;
    const int plane = Hp * Wp;
    const int TL = kBQ + 2 * Wp + 2;
    const int NIN = (TL + 63) >> 6;                          // 1-KiB pieces of one input plane of a chunk (<= 16, checked at launch)
    const int INU = NIN * 64;                                // 16-byte units

    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // persistent walk: workgroup id -> (id % 8) owns a contiguous slice of the logical tile list (the dispatcher deals ids
    // round-robin over the 8 XCDs), and the workgroups of one XCD step through it together with stride nx: at any moment they
    // work on neighbouring tiles, whose 2 * Wp + 2 halos meet in that XCD's L2.  With grid == ntiles this is tile_index().
    const int P = gridDim.x, xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    const int nx = (P >> 3) + (xcd < (P & 7) ? 1 : 0);                           // workgroups on this XCD
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int tcnt = per + (xcd < rem ? 1 : 0), tstart = xcd * per + (xcd < rem ? xcd : rem);
    if (wslot >= tcnt) return;
    // The two workgroups that share a CU start together and their tiles take the same time: left alone they stay in LOCK STEP for
    // the whole walk -- both inside the chunk loop (sharing the matrix pipe), then both outside it (pipe idle).  Persistent
    // workgroups keep whatever phase offset they start with, so the one in the odd wave slot starts `desync` x 8128 cycles late:
    // its chunk loop then runs under the other's epilogue / barriers / copy issue and vice versa.
    if (desync > 0 && wslot + nx < tcnt && (__builtin_amdgcn_s_getreg((4 << 11) | 4) & 1u))       // HW_REG_HW_ID[3:0] = wave slot
        for (int i = 0; i < desync; ++i) __builtin_amdgcn_s_sleep(127);
    const int C8_0 = args()->seg[0].C8;
    const int nchunk = args()->nchunk;

    // DMA-source state of the tile whose chunks are being STAGED (the current tile, or -- during a tile's last chunk -- the
    // workgroup's next tile) and epilogue state of the tile being MULTIPLIED; both live in SGPRs
    struct Src {
        int bq, cb, n;
        const uint4 *hi0, *lo0, *hi1, *lo1;   // this sequence's first plane in segment 0 / 1 (hi and lo tensors)
        const uint4* wsrc;                    // weight images of (set, cout block)
    };
    struct Ep { int bq, cb, n; const float* aux; };
    constexpr int kMaxIn = 4;                                // input pieces per wave: NIN <= 16 (Wp <= 254)
    constexpr int kMaxW = (2 * WPIECES + 3) / 4;             // weight pieces per wave (TERMS == 3: hi | lo planes)
    auto src_of = [&](int tk) {
        // (the divisors go through `opaque`: otherwise hipcc hoists three sets of reciprocal constants out of the tile walk
        // and keeps them in SGPRs for the whole kernel)
        const int lid = tstart + tk;
        Src t;
        const int dq = opaque(nblk_q), dc = opaque(ncb);
        t.bq = lid % dq;
        const int rest = lid / dq;
        t.cb = rest % dc; t.n = rest / dc;
        const KArgs ka = args();
        const int nps = opaque(ka->c.n_per_set);
        const int set = t.n / nps, nn = t.n - set * nps;
        const long off0 = (long)nn * ka->seg[0].stride_n + ka->seg[0].set_off[set];
        const long off1 = (long)nn * ka->seg[1].stride_n + ka->seg[1].set_off[set];
        t.hi0 = ka->seg[0].hi + off0; t.lo0 = ka->seg[0].lo + off0;
        t.hi1 = ka->seg[1].hi + off1; t.lo1 = ka->seg[1].lo + off1;
        const int npack = ka->nchunk_pack > 0 ? ka->nchunk_pack : nchunk;
        t.wsrc = ka->w + (long)set * ka->w_set_stride + (long)t.cb * npack * (2 * WUNITS);
        return t;
    };
    auto ep_of = [&](const Src& t) {
        const KArgs ka = args();
        const int set = t.n / opaque(ka->c.n_per_set);
        const float* ax = ka->c.aux;
        return Ep{t.bq, t.cb, t.n, ax ? ax + (long)set * ka->c.aux_set_stride : nullptr};
    };
    auto dma = [&](const char* g, int lds_unit) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + lds_unit), 16, 0, 0);
    };
    auto in_plane = [&](const Src& t, int c, bool lo_plane) -> const uint4* {
        const bool first = c < C8_0;
        const uint4* base = first ? (lo_plane ? t.lo0 : t.hi0) : (lo_plane ? t.lo1 : t.hi1);
        return base + (long)(first ? c : c - C8_0) * plane;
    };
    // Copies are issued as straight-line steps at fixed K blocks of the running chunk (a generic "next n pieces" loop costs a
    // wave tens of scalar branches per piece, during which it issues no MFMAs): this wave's share = input pieces wave,
    // wave + 4, ... and weight pieces likewise.  Per-lane source position of an input piece p: q0 + 64 * p + lane, clamped
    // into the plane (positions past the end only feed outputs that the epilogue drops).
    auto issue_in = [&](const Src& t, const uint4* src, int in_unit) {
        if (!src) return;
        const char* sp = reinterpret_cast<const char*>(src);
        const int qb = t.bq * kBQ + lane;
        const int wv = opaque(wave);         // keeps the per-piece predicates / LDS offsets from being hoisted into SGPRs for the whole kernel
#pragma unroll
        for (int k = 0; k < kMaxIn; ++k) {
            const int pid = wv + 4 * k;
            if (pid < NIN) {
                int q = qb + 64 * pid;
                q = q < plane ? q : plane - 1;
                dma(sp + (unsigned)q * 16u, in_unit + 64 * pid);
            }
        }
    };
    auto issue_w = [&](const uint4* ws, int w_unit, int nwp, int k0, int k1) {      // this wave's weight pieces k0 .. k1 - 1
        if (!ws) return;
        const char* wp8 = reinterpret_cast<const char*>(ws);
        const int wv = opaque(wave);
#pragma unroll
        for (int k = 0; k < kMaxW; ++k) {
            const int wp = wv + 4 * k;
            if (k >= k0 && k < k1 && wp < nwp) dma(wp8 + (unsigned)((64 * wp + lane) * 16), w_unit + 64 * wp);
        }
    };

    // operand slots (16-byte units) of K block kb: tap = 2 * kb + hi (clamped to 8), B at wave * 128 + lo + (tap / 3) * Wp + tap % 3,
    // A at tap * BN + lo.  They are formed where they are used from two base registers (ten persistent slot registers were what
    // pushed the 64-cout kernels over 256 VGPRs once the tile loop kept them alive across the epilogue); `hsel` is re-laundered
    // per product so that the compiler does not hoist the ten sums back out.
    const int bbase = wave * (kQG * 32) + lo, abase = lo;
    auto bslot_of = [&](int kb, int hsel) {
        const int t0 = 2 * kb, t1 = (2 * kb + 1) > 8 ? 8 : 2 * kb + 1;
        const int o0 = (t0 / 3) * Wp + (t0 % 3), o1 = (t1 / 3) * Wp + (t1 % 3);
        return bbase + (hsel ? o1 : o0);
    };
    auto aslot_of = [&](int kb, int hsel) {
        const int t0 = 2 * kb, t1 = (2 * kb + 1) > 8 ? 8 : 2 * kb + 1;
        return abase + (hsel ? t1 * BN : t0 * BN);
    };
    auto vopaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    const bool zero_half = hi != 0;            // K block 4: the second half-wave has no tap

    f32x16 acc[NCG][kQG];
    // One product of a chunk: x(in_unit) * w(w_unit) over its 5 K blocks.  Operands of K block kb + 1 are fetched before
    // the MFMAs of kb issue; `step(kb)` runs after the MFMAs of K block kb (copy issue).
    auto load_ab = [&](int kb, int hsel, int in_unit, int w_unit, v8 (&av)[NCG], v8 (&bv)[kQG]) {
        const int as = w_unit + aslot_of(kb, hsel), bs = in_unit + bslot_of(kb, hsel);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            av[g] = *reinterpret_cast<const v8*>(smem + as + g * 32);
            if (kb == kKB - 1 && zero_half) av[g] = v8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < kQG; ++j) bv[j] = *reinterpret_cast<const v8*>(smem + bs + 32 * j);
    };
    auto mfma_chunk_s = [&](int in_unit, int w_unit, auto&& step) {
        v8 av[2][NCG], bv[2][kQG];
        const int hsel = vopaque(hi);
        load_ab(0, hsel, in_unit, w_unit, av[0], bv[0]);
#pragma unroll
        for (int kb = 0; kb < kKB; ++kb) {
            if (kb + 1 < kKB) load_ab(kb + 1, hsel, in_unit, w_unit, av[(kb + 1) & 1], bv[(kb + 1) & 1]);
#pragma unroll
            for (int g = 0; g < NCG; ++g)
#pragma unroll
                for (int j = 0; j < kQG; ++j) acc[g][j] = E::mfma(av[kb & 1][g], bv[kb & 1][j], acc[g][j]);
            step(kb);
        }
    };
    // the two products that read the hi tile: x_hi * w_lo + x_hi * w_hi (B operands fetched once)
    auto load_a2b = [&](int kb, int hsel, int in_unit, int w_unit, v8 (&ah)[NCG], v8 (&al)[NCG], v8 (&bv)[kQG]) {
        const int as = w_unit + aslot_of(kb, hsel), bs = in_unit + bslot_of(kb, hsel);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            ah[g] = *reinterpret_cast<const v8*>(smem + as + g * 32);
            al[g] = *reinterpret_cast<const v8*>(smem + as + WUNITS + g * 32);
            if (kb == kKB - 1 && zero_half) { ah[g] = v8{0, 0, 0, 0, 0, 0, 0, 0}; al[g] = v8{0, 0, 0, 0, 0, 0, 0, 0}; }
        }
#pragma unroll
        for (int j = 0; j < kQG; ++j) bv[j] = *reinterpret_cast<const v8*>(smem + bs + 32 * j);
    };
    // nkb = 5: all tap pairs (the last one half empty); nkb = 4: taps 0..7 only -- tap 8 of this chunk shares a K block with
    // tap 8 of its pair chunk (`straddle`)
    auto mfma_chunk_hi2_s = [&](int in_unit, int w_unit, int nkb, auto&& step) {
        v8 ah[2][NCG], al[2][NCG], bv[2][kQG];
        const int hsel = vopaque(hi);
        load_a2b(0, hsel, in_unit, w_unit, ah[0], al[0], bv[0]);
#pragma unroll
        for (int kb = 0; kb < kKB; ++kb) {
            if (kb + 1 < kKB - 1 || (kb + 1 == kKB - 1 && nkb == kKB))
                load_a2b(kb + 1, hsel, in_unit, w_unit, ah[(kb + 1) & 1], al[(kb + 1) & 1], bv[(kb + 1) & 1]);
            if (kb < kKB - 1 || nkb == kKB) {
#pragma unroll
                for (int g = 0; g < NCG; ++g)
#pragma unroll
                    for (int j = 0; j < kQG; ++j) acc[g][j] = E::mfma(al[kb & 1][g], bv[kb & 1][j], acc[g][j]);
#pragma unroll
                for (int g = 0; g < NCG; ++g)
#pragma unroll
                    for (int j = 0; j < kQG; ++j) acc[g][j] = E::mfma(ah[kb & 1][g], bv[kb & 1][j], acc[g][j]);
            }
            step(kb);
        }
    };
    // The K block that removes the tenth tap half from the two hi-tile products of a PAIR of chunks: half-wave 0 multiplies tap 8
    // of the chunk in buffer parity b, half-wave 1 tap 8 of the chunk in parity b ^ 1 (both hi tiles and both weight images are
    // resident: they are double-buffered).  x_hi * (w_lo, w_hi).
    auto straddle = [&](int b, int inu, int wu0, int wstride) {
        const int hsel = vopaque(hi);
        const int bb = hsel ? (b ^ 1) : b;
        const int as = wu0 + bb * wstride + abase + 8 * BN, bs = bb * inu + bbase + 2 * Wp + 2;
        v8 ah[NCG], al[NCG], bv[kQG];
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            ah[g] = *reinterpret_cast<const v8*>(smem + as + g * 32);
            al[g] = *reinterpret_cast<const v8*>(smem + as + WUNITS + g * 32);
        }
#pragma unroll
        for (int j = 0; j < kQG; ++j) bv[j] = *reinterpret_cast<const v8*>(smem + bs + 32 * j);
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int j = 0; j < kQG; ++j) acc[g][j] = E::mfma(al[g], bv[j], acc[g][j]);
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int j = 0; j < kQG; ++j) acc[g][j] = E::mfma(ah[g], bv[j], acc[g][j]);
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int j = 0; j < kQG; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][j][r] = 0.0f;
    };
    auto epilogue = [&](const Ep& t) {
        H16Args ea;                                          // only the fields the epilogue reads are actually loaded
        __builtin_memcpy(&ea, args(), sizeof(H16Args));
        if constexpr (OUT == OUT_B16) h16_epilogue_b16<BF, NCG, EPI>(ea, acc, t.n, t.cb, t.bq, nblk_q, t.aux, tid);
        else conv_epilogue<NCG, EPI>(ea.c, acc, t.n, t.cb, t.bq, nblk_q, t.aux, tid);
    };

    if constexpr (TERMS == 1) {
        // LDS: 3 stages of [input tile | weight plane hi]; per tile, no prefetch across tiles (the opt-in plain-16-bit path)
        const int STU = INU + WUNITS;
        const int cnt_in = (NIN + 3) >> 2;
        auto issue_stage = [&](const Src& t, int c, int stage) {       // every wave issues exactly `cnt` copies (a surplus repeats the last piece)
            const char* sp = reinterpret_cast<const char*>(in_plane(t, c, false));
            const char* wp8 = reinterpret_cast<const char*>(t.wsrc + (long)c * (2 * WUNITS));
            const int qb = t.bq * kBQ + lane;
            const int wv = opaque(wave);
#pragma unroll
            for (int k = 0; k < kMaxIn; ++k)
                if (k < cnt_in) {
                    int pid = wv + 4 * k;
                    pid = pid < NIN ? pid : NIN - 1;
                    int q = qb + 64 * pid;
                    q = q < plane ? q : plane - 1;
                    dma(sp + (unsigned)q * 16u, stage * STU + 64 * pid);
                }
#pragma unroll
            for (int k = 0; k < (WPIECES + 3) / 4; ++k) {
                int wp = wv + 4 * k;
                wp = wp < WPIECES ? wp : WPIECES - 1;
                dma(wp8 + (unsigned)((64 * wp + lane) * 16), stage * STU + INU + 64 * wp);
            }
        };
        for (int tk = wslot; tk < tcnt; tk += nx) {
            const Src t = src_of(tk);
            if (tk != wslot) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cbarrier(); }   // every wave has left the previous tile's LDS
            zero_acc();
            issue_stage(t, 0, 0);
            if (nchunk > 1) issue_stage(t, 1, 1);
            int st = 0;                                     // stage of chunk c
            for (int c = 0; c < nchunk; ++c) {
                // chunk c has landed.  (Round 2 waited with a COUNTED vmcnt so that chunk c + 1 stayed in flight; inside the tile
                // walk the compiler may place a scratch access between the copies and the wait, which would shift the count
                // onto a copy of THIS chunk -- chunk c + 1 was issued a whole chunk ago, so waiting for it costs little.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                cbarrier();                                 // ... for every wave; and every wave is done reading chunk c - 1
                const int s2 = st == 0 ? 2 : st - 1;        // (c + 2) % 3 == (c - 1) % 3
                const bool pre = c + 2 < nchunk;
                mfma_chunk_s(st * STU, st * STU + INU, [&](int kb) { if (kb == 0 && pre) issue_stage(t, c + 2, s2); });
                st = st == 2 ? 0 : st + 1;
            }
            epilogue(ep_of(t));
        }
    } else {
        // LDS: HI[0] HI[1] LO W[0] W[1], W = [hi plane | lo plane].  The chunk stream runs ACROSS tiles: buffer parity follows the
        // global chunk counter g, and the stage after the last chunk of a tile is chunk 0 of the workgroup's next tile -- `src`
        // always describes the tile whose chunks are being STAGED, `ep` the tile being multiplied.
        // TERMS == 2 (ttc_config.two_term_layers): x_hi * (w_lo, w_hi) only -- 16-bit activations, exact weights.  No lo tile is staged
        // (LDS: HI[0] HI[1] W[0] W[1]) and a chunk has ONE barrier.
        const int LOU = 2 * INU, WU0 = TERMS == 3 ? 3 * INU : 2 * INU;
        Src src = src_of(wslot);
        Ep ep = ep_of(src);
        issue_in(src, in_plane(src, 0, false), 0);
        issue_w(src.wsrc, WU0, 2 * WPIECES, 0, kMaxW);
        if constexpr (TERMS == 3) issue_in(src, in_plane(src, 0, true), LOU);
        // one chunk of the stream: multiply the chunk in buffer parity g & 1 while (cn, any) -- chunk cn of `src`, if any -- is staged.
        // mode 0: a chunk on its own (5 + 5 K blocks).  Chunks 2i and 2i + 1 of a tile form a PAIR whose hi-tile products skip the
        // half-empty fifth K block: mode 1 (first of the pair) ends with the straddle block over both chunks' tap 8 -- by then the
        // second chunk's hi tile and weights have landed, which is exactly the wait the second chunk would start with -- and mode 2
        // (second) runs taps 0..7 only: 28 K-block products per pair instead of 30.
        auto chunk = [&](int g, int cn, bool any, int mode) {
            const int b = g & 1;
            const uint4* nhi = any ? in_plane(src, cn, false) : nullptr;
            const uint4* nlo = any ? in_plane(src, cn, true) : nullptr;
            const uint4* nws = any ? src.wsrc + (long)cn * (2 * WUNITS) : nullptr;
            const int nwu = WU0 + (b ^ 1) * 2 * WUNITS;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // hi, lo and weights of this chunk have landed (and, after an
            cbarrier();                                               // epilogue, its stores have retired) ... for every wave; every
                                                                      // wave is done with the previous chunk
            auto stage_next = [&](int kb) {                           // next hi tile after K block 0, weights after 1 and 2
                if (kb == 0) issue_in(src, nhi, (b ^ 1) * INU);
                else if (kb == 1) issue_w(nws, nwu, 2 * WPIECES, 0, (kMaxW + 1) / 2);
                else if (kb == 2) issue_w(nws, nwu, 2 * WPIECES, (kMaxW + 1) / 2, kMaxW);
            };
            if constexpr (TERMS == 3) {
                mfma_chunk_s(LOU, WU0 + b * 2 * WUNITS, stage_next);   // x_lo * w_hi
                cbarrier();                                           // the lo tile is free
                mfma_chunk_hi2_s(b * INU, WU0 + b * 2 * WUNITS, mode == 0 ? kKB : kKB - 1,
                                 [&](int kb) { if (kb == 0) issue_in(src, nlo, LOU); });   // x_hi * (w_lo, w_hi)
            } else {
                (void)nlo;
                mfma_chunk_hi2_s(b * INU, WU0 + b * 2 * WUNITS, mode == 0 ? kKB : kKB - 1, stage_next);   // x_hi * (w_lo, w_hi)
            }
            if (mode == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the pair chunk's tiles (issued during this chunk) have landed
                cbarrier();
                straddle(b, INU, WU0, 2 * WUNITS);
            }
        };
        auto mode_of = [&](int c) { return (c & 1) ? 2 : (c + 1 < nchunk ? 1 : 0); };
        int g = 0;
        // probe aid (ttc_debug_knob 2 / 3, tools/probes/h16_trace.py): thread 0 stamps s_memtime at the phase boundaries of the
        // workgroup's first 12 tiles -- [tile][0] start, [1] before the last chunk, [2] after it, [3] after the epilogue; slot 63 = HW ids
        unsigned long long* tr = (TRACE && trace && tid == 0) ? trace + (long)blockIdx.x * 64 : nullptr;
        if (TRACE && tr) {
            tr[63] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | __builtin_amdgcn_s_getreg((3 << 11) | 20);
            tr[60] = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz reference clock: with [62] / the end stamps = the shader clock
            tr[62] = __builtin_amdgcn_s_memtime();
        }
        int ti = 0;
        for (int tk = wslot;; ++ti) {
            if (TRACE && tr && ti < 12) tr[4 * ti] = __builtin_amdgcn_s_memtime();
            zero_acc();
            for (int c = 0; c + 1 < nchunk; ++c, ++g) chunk(g, c + 1, true, mode_of(c));
            const bool any = tk + nx < tcnt;                          // last chunk of this tile: stage chunk 0 of the next one
            if (any) src = src_of(tk + nx);
            if (TRACE && tr && ti < 12) tr[4 * ti + 1] = __builtin_amdgcn_s_memtime();
            chunk(g, 0, any, (nchunk & 1) ? 0 : 2);
            ++g;
            if (TRACE && tr && ti < 12) tr[4 * ti + 2] = __builtin_amdgcn_s_memtime();
            epilogue(ep);                                             // no LDS: the next tile's first chunk is landing meanwhile
            if (TRACE && tr && ti < 12) tr[4 * ti + 3] = __builtin_amdgcn_s_memtime();
            if (TRACE && tr && !any) { tr[61] = __builtin_amdgcn_s_memrealtime(); tr[59] = __builtin_amdgcn_s_memtime(); }
            if (!any) break;
            tk += nx;
            ep = ep_of(src);
        }
    }
}

// probe knobs (ttc_debug_knob): [0] persistent grid size (-1 = 2 x CUs, 0 = one workgroup per tile), [1] start offset of the odd
// wave slot in units of s_sleep(127) (-1 = default), [2] | [3] low / high half of a device pointer to a trace buffer, [4] the
// epilogue kind to trace
int g_h16_knob[8] = {-1, -1, -1, -1, -1, -1, -1, -1};

// 2 workgroups per CU of the CURRENT device (cached per device id; TTC_H16_PERSIST overrides for probes)
int resident_workgroups() {
    static const int forced = [] { const char* e = getenv("TTC_H16_PERSIST"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced;
    static std::atomic<int> per_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
    int v = per_dev[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        v = 2 * cus;
        per_dev[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
constexpr int kDesyncDefault = 0;

template <int BF, int TERMS, int NCG, int EPI, int OUT>
hipError_t launch_h16(const H16Args& a, const PackedConv& pw, int n, hipStream_t s) {
    constexpr int BN = NCG * 32;
    constexpr int WPIECES = (9 * BN * 16 + 1023) / 1024;
    const int TL = kBQ + 2 * a.c.Wp + 2;
    const int NIN = (TL + 63) >> 6;
    const size_t lds = TERMS == 1 ? (size_t)3 * (NIN + WPIECES) * 1024 : (size_t)((TERMS == 3 ? 3 : 2) * NIN + 4 * WPIECES) * 1024;
    if (lds > 160 * 1024 || NIN > 16) return hipErrorInvalidValue;
    static LdsConfig lds_cfg;
    if (hipError_t e = lds_cfg.ensure(&conv3x3_h16<BF, TERMS, NCG, EPI, OUT>, lds); e != hipSuccess) return e;
    const int nblk_q = conv_q_blocks(a.c.Hp, a.c.Wp);
    const int ntiles = nblk_q * pw.ncb * n;
    // persistent grid = the resident set: 2 workgroups per CU (TTC_H16_PERSIST overrides; 0 = one workgroup per tile)
    const int resident = resident_workgroups();
    const int res = g_h16_knob[0] >= 0 ? g_h16_knob[0] : resident;
    const int grid = res > 0 ? std::min(ntiles, res) : ntiles;
    const int desync = g_h16_knob[1] >= 0 ? g_h16_knob[1] : kDesyncDefault;
    // trace buffer (device pointer in knobs 2 | 3, 64 x u64 per workgroup) for the layer kind in knob 4 (default: the ConvGRU gates)
    unsigned long long* trace = nullptr;
    if (g_h16_knob[2] != -1 || g_h16_knob[3] != -1) {
        const int want_epi = g_h16_knob[4] >= 0 ? g_h16_knob[4] : (int)EPI_RAW;
        if (EPI == want_epi && TERMS == 3)
            trace = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)g_h16_knob[3] << 32) | (unsigned)g_h16_knob[2]);
    }
    if constexpr (TERMS == 3 && EPI <= EPI_SWISH && BF == 0) {        // the traced instantiations exist for the fp16 GroupNorm layers only
        if (trace) {
            static LdsConfig lds_tr;
            if (hipError_t e = lds_tr.ensure(&conv3x3_h16<BF, TERMS, NCG, EPI, OUT, true>, lds); e != hipSuccess) return e;
            hipLaunchKernelGGL((conv3x3_h16<BF, TERMS, NCG, EPI, OUT, true>), dim3(grid), dim3(kThreads), lds, s, a, nblk_q, pw.ncb, ntiles, desync, trace);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((conv3x3_h16<BF, TERMS, NCG, EPI, OUT>), dim3(grid), dim3(kThreads), lds, s, a, nblk_q, pw.ncb, ntiles, desync, nullptr);
    return hipGetLastError();
}

}  // namespace

void h16_set_knob(int which, int value) { if (which >= 0 && which < 8) g_h16_knob[which] = value; }

// ---- host: 16-bit conversions (round to nearest even), weight packing, dispatch -----------------------------------
uint16_t h16_from_float(float f, bool bf) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if (bf) {
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                        // NaN
    if (ax >= 0x477ff000u) return (uint16_t)(sign | (ax >= 0x7f800000u ? 0x7c00u : 0x7bffu));   // inf stays inf; finite overflow saturates
    if (ax < 0x33000001u) return (uint16_t)sign;                                    // < 2^-25: rounds to zero
    const int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;                                                                      // bits to drop from the 24-bit mantissa
    uint32_t base;
    if (e >= -14) { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }  // normal: drop the hidden bit into the exponent
    else { shift = 13 + (-14 - e); base = 0; }                                      // subnormal
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | (base + r));                                           // a mantissa carry bumps the exponent correctly
}

float h16_to_float(uint16_t h, bool bf) {
    uint32_t u;
    if (bf) u = (uint32_t)h << 16;
    else {
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
        if (e == 0) {
            if (m == 0) u = sign;
            else {                                                                  // subnormal: value = m * 2^-24
                float f = (float)m * 5.9604644775390625e-08f;
                std::memcpy(&u, &f, 4);
                u |= sign;
            }
        } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
        else u = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// LDS images per (set, cout block, chunk): [hi | lo] planes of [tap 0..8][cout 0..BN)[8 ch] 16-bit, each plane padded to
// whole 1-KiB DMA pieces.  Chunks follow the blocked input layout: the first segment's C0 channels are padded to a multiple
// of 8 on their own, the second segment starts on a block boundary.  Returns 16-byte units per set.
long conv_pack_h16(const float* const* hwio, int nsets, int Cin, int C0, int Cout, int BN, bool bf, std::vector<uint16_t>& out,
                   int* nchunk_out) {
    const int c8_0 = (C0 + 7) / 8, c8_1 = (Cin - C0 + 7) / 8, nchunk = c8_0 + c8_1, ncb = (Cout + BN - 1) / BN;
    const int wpieces = (9 * BN * 16 + 1023) / 1024;
    const long plane_el = (long)wpieces * 512;                              // u16 elements per padded plane
    const long per_set = (long)ncb * nchunk * 2 * plane_el;
    out.assign((size_t)per_set * nsets, 0);
    for (int s = 0; s < nsets; ++s)
        for (int cb = 0; cb < ncb; ++cb)
            for (int c = 0; c < nchunk; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    for (int co = 0; co < BN; ++co)
                        for (int k = 0; k < 8; ++k) {
                            const int ci = c < c8_0 ? c * 8 + k : C0 + (c - c8_0) * 8 + k;
                            const bool real = c < c8_0 ? (c * 8 + k < C0) : (ci < Cin);
                            const int o = cb * BN + co;
                            if (!real || o >= Cout) continue;
                            const float w = hwio[s][((long)tap * Cin + ci) * Cout + o];       // HWIO, tap = 3*dy + dx
                            const uint16_t h = h16_from_float(w, bf), l = h16_from_float(w - h16_to_float(h, bf), bf);
                            const long base = (long)s * per_set + (((long)cb * nchunk + c) * 2) * plane_el;
                            const long e = ((long)tap * BN + co) * 8 + k;
                            out[base + e] = h;
                            out[base + plane_el + e] = l;
                        }
    if (nchunk_out) *nchunk_out = nchunk;
    return per_set / 8;
}


// matrix-instruction flops of a launch: tile = 512 flat positions x BN couts; a K-block product (K = 16: the 8 channels of two taps) is
// 2 * BN * 512 * 16 flops.  TERMS = 3: 28 K-block products per chunk PAIR (tap 8 of the two hi-tile products shares a block), 15 for a
// single chunk; TERMS = 2 (the two hi-tile products): 18 / 10; TERMS = 1: 5 per chunk.
double conv_issued_flops_h16(const H16Args& a, const PackedConv& pw, int n) {
    const double tiles = (double)conv_q_blocks(a.c.Hp, a.c.Wp) * pw.ncb * n;
    const int pairs = a.nchunk / 2, single = a.nchunk & 1;
    const double kprod = pw.terms == 1 ? 5.0 * a.nchunk : (pw.terms == 2 ? 18.0 * pairs + 10.0 * single : 28.0 * pairs + 15.0 * single);
    return tiles * kprod * 2.0 * pw.BN * kBQ * 16.0;
}

hipError_t conv_launch_h16(const H16Args& a, const PackedConv& pw, int mode, int epi, int out_kind, int n, hipStream_t s) {
    const bool bf = mode == 1;
    const int terms = pw.terms == 1 ? 1 : (pw.terms == 2 ? 2 : 3);
#define TTC_H16_CASE(BFV, T, ncg, e, o) \
    if ((int)bf == BFV && terms == T && pw.BN == ncg * 32 && epi == e && out_kind == o) return launch_h16<BFV, T, ncg, e, o>(a, pw, n, s);
#define TTC_H16_LAYERS(BFV, T)                                                                                       \
    TTC_H16_CASE(BFV, T, 2, EPI_RAW, OUT_B16)             /* ConvGRU gates           -> raw fp32, blocked halves  */ \
    TTC_H16_CASE(BFV, T, 1, EPI_SSE, OUT_B16)             /* ConvGRU candidate       -> raw fp32, blocked halves  */ \
    TTC_H16_CASE(BFV, T, 2, EPI_SWISH, OUT_B16)           /* conv_swish_gn blocks    -> raw fp32, blocked halves  */ \
    TTC_H16_CASE(BFV, T, 1, EPI_BIAS_RELU, OUT_B16)       /* DSen2 in / x1 convs     -> blocked 16-bit pair       */ \
    TTC_H16_CASE(BFV, T, 1, EPI_BIAS_RES, OUT_B16)        /* DSen2 residual convs    -> blocked 16-bit pair       */ \
    TTC_H16_CASE(BFV, T, 1, EPI_BIAS_TANH_ADD, OUT_F32)   /* DSen2 head              -> fp32 planar               */
    TTC_H16_LAYERS(0, 3)
    TTC_H16_LAYERS(0, 1)
    TTC_H16_LAYERS(1, 3)
    // two products (x_hi * w): the GroupNorm layers of the fp16 engine only (ttc_config.two_term_layers)
    TTC_H16_CASE(0, 2, 2, EPI_RAW, OUT_B16)
    TTC_H16_CASE(0, 2, 1, EPI_SSE, OUT_B16)
    TTC_H16_CASE(0, 2, 2, EPI_SWISH, OUT_B16)
#undef TTC_H16_LAYERS
#undef TTC_H16_CASE
    return hipErrorInvalidValue;
}
