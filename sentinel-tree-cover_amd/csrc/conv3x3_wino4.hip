// Winograd F(4x4, 3x3) form of the 3x3 convolution on the gfx950 fp32 matrix instruction v_mfma_f32_16x16x4_f32 for the 64-cout-multiple
// GroupNorm layers of the fp32 engine (ConvGRU gates, the conv_swish_gn blocks: src/train/src/model.py:251, :416-442, :448-538).
//
//     Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A          d: 6 x 6 input patch, g: 3 x 3 kernel, Y: 4 x 4 outputs
//
// 36 independent GEMMs  M_xi[cout][tile] = sum_cin U_xi[cout][cin] * V_xi[cin][tile]  (xi = 6 a + b): 36 multiply-accumulates per 16
// outputs = 2.25 per (output, cin, cout) against 4 for F(2x2, 3x3) (conv3x3_wino.hip) and 9 for the direct form.
//
// What round 5 measured first (tools/probes/mfma_f32_filler_probe.hip, mfma_f32_16x16_order_probe.hip) and what the design follows from:
//   * the fp32 matrix instruction runs on the vector ALU's own FMA lanes: a VALU instruction NEVER overlaps an fp32 MFMA of its own wave
//     or of the SIMD's other wave -- each costs ~4 cycles (6 for a lone wave) on top of the matrix stream.  v_pk_*_f32 costs the same as
//     the scalar form, so every transform below is written on float2 pairs: half the ALU time for the same arithmetic;
//   * ds_read / scalar / (few) global loads DO hide under the SIMD partner's ALU work, ds_write costs ~5 cycles: two waves per SIMD, not
//     one fat wave (a first version with one 512-register wave per SIMD and 288 accumulators measured 8.3-9.7 k cycles per 4.6 k-cycle
//     chunk: nothing of a lone wave's stream overlaps);
//   * two waves per SIMD issue the 16x16x4 form every 24.9 cycles (one wave alone: 32.1);
//   * more than 256 accumulator registers per wave make hipcc shuttle the excess through v_accvgpr moves (43-46 cycles per MFMA).
//   * an 8-byte global load per (xi, lane) is ~14 cycles of the CU's ONE vector-memory path per wave instruction and LDS-DMA lands a
//     piece only every ~90 cycles: a version in which both waves of a SIMD fetched all 36 A operands was bound by that path, one that
//     brought U and the inputs in by LDS-DMA by the DMA landing rate (experiments/README.md, round 5).
// Mapping.  Workgroup = 8 waves = 2 per SIMD, one workgroup per CU, tile = 64 couts x 32 tiles (two independent 4 x 4-tile sub-regions =
// 2 x 16 x 16 output pixels).  The two waves of a SIMD share a cout block (16 couts) and SPLIT THE xi BY ROWS: wave w holds rows
// a = 3 h .. 3 h + 2 (h = w >> 2; 18 xi) for BOTH tile blocks: 18 x 2 x 4 = 144 accumulator registers.  Every A operand (one 8-byte
// global load per (xi, lane) and chunk, a full chunk ahead) and every B operand (ONE ds_read_b128 per xi: both tile blocks, both k-steps)
// feeds four matrix instructions, and no operand is fetched twice by a SIMD: half the loads per MFMA of any split by tile block.
//   * output transform: each wave applies the row half it can (Z[a][j] = sum_b M[a][b] A[b][j], its three rows a), sends the Z of the
//     OTHER tile block to its partner through LDS (the idle V buffer, the idle staged image and 42 KB kept for it: 2 barriers per tile)
//     and finishes A^T Z, the epilogue op, GroupNorm partial sums and stores of its own tile block;
//   * V = B^T d B is computed once per workgroup by waves 0..3 (one HALF patch of a channel pair (c, c + 4) per lane: packed arithmetic
//     over the pair, which is also the pair of k-steps a B operand holds);
//   * waves 4..7 stage the next chunks' inputs (global -> registers -> LDS, channel pairs interleaved so that a patch element is one
//     8-byte LDS word): scalar base + per-tile lane offsets, no vector address arithmetic in the chunk loop;
//   * ONE barrier per 8-channel chunk (staged image and V are both double-buffered); the roles are separate straight-line instantiations
//     of the whole tile walk (a branch around a load makes hipcc wait vmcnt(0) at every join);
//   * persistent walk with the chunk stream running ACROSS tiles, tile ids decomposed by multiply-shift on the scalar unit, kernel
//     arguments re-read from the kernarg segment -- the devices of conv3x3_wino.hip.
// Transform constants (Lavin & Gray): B^T rows (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1);
// A^T rows (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,0) (0,1,-1,8,-8,1); G in conv_pack_wino4 (double).  fp32 throughout.
#include <algorithm>
#include <type_traits>

#include "conv_common.h"

using namespace ttcconv;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kCK = 8;                   // input channels per chunk = 2 k-steps of 4
constexpr int kIR = 18;                  // staged rows / columns of a sub-region: 4 tiles x 4 + 2
constexpr int kIP = 20;                  // staged row pitch in ELEMENTS (an element = the float2 of channels (c, c + 4))
constexpr int kKQ = kIR * kIP * 2;       // floats of one channel-pair plane of a sub-region image (720)
constexpr int kSUB = 4 * kKQ + 32;       // floats of one sub-region image
constexpr int kIN = 2 * kSUB;            // floats of one staged chunk image (two sub-regions)
constexpr int kSE = kIR * (kIR / 2);     // float2 (two pixels of one channel) per channel of a sub-region image (162)
constexpr int kVB = 36 * 256;            // floats of one V buffer: [xi 36][kq 4][tile 16][tb 2][s 2]
#ifndef TTC_W4_RING
#define TTC_W4_RING 6
#endif
constexpr int kAR = TTC_W4_RING;         // A operands requested ahead (of the wave's 18 xi per chunk); 18 % kAR == 0
constexpr int kXS = 64 * 24;             // floats of one exchange half-slot: 24 per lane (two couts x three rows a x four columns j)
constexpr int kThreads4 = 512;

struct Wino4Args {
    ConvArgs a;
    const float* U; long u_set_stride;   // [set][cout block 16][chunk][xi 36][kq 4][cout 16][s 2]: channel 8 c + 4 s + kq
    int nchunk, nks_last, RXn, RYn, ncq, ntiles;
    int RR, S, pps;                      // sub-regions per window, per weight set, sub-region PAIRS per weight set
    unsigned long long m_cq, m_pps, m_rr, m_rx;   // floor(2^40 / d) + 1
    int nrun, cin_run;
    unsigned long long* trace;           // TTC_WINO4_TRACE: 8 x u64 per workgroup (phase sums, tiles, s_memrealtime / s_memtime at both ends)
    int probe;                           // ablation bits (TTC_WINO4_PROBE, timing only): 1 no output stores, 2 no input transform, 4 no epilogue,
                                         // 8 no staging loads, 16 no A-operand refills, 32 no B-operand reads, 64 no chunk barrier
};

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ v2f pk_fma(float k, v2f x, v2f y) { return __builtin_elementwise_fma(v2f{k, k}, x, y); }

// the six-point 1-D input transform B^T on packed pairs (14 packed instructions)
__device__ __forceinline__ void bt6(const v2f (&d)[6], v2f (&o)[6]) {
    const v2f s12 = d[1] + d[2], m12 = d[1] - d[2], s34 = d[3] + d[4], m43 = d[4] - d[3], m31 = d[3] - d[1], m42 = d[4] - d[2];
    o[0] = pk_fma(4.f, d[0], pk_fma(-5.f, d[2], d[4]));
    o[1] = pk_fma(-4.f, s12, s34);
    o[2] = pk_fma(4.f, m12, m43);
    o[3] = pk_fma(2.f, m31, m42);
    o[4] = pk_fma(-2.f, m31, m42);
    o[5] = pk_fma(4.f, d[1], pk_fma(-5.f, d[3], d[5]));
}

template <int EPI, int PROBE = 0>
__global__ __launch_bounds__(kThreads4, 2) void conv3x3_wino4(Wino4Args wa) {
    typedef const __attribute__((address_space(4))) Wino4Args* KArgs;
    const KArgs kp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    auto args = [&]() { KArgs q = kp; asm volatile("" : "+s"(q)); return q; };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_tile = smem;                       // [2 buffers][2 sub-regions][4 kq][18 rows][20 cols][2 s] (+ pad)
    float* Vb = smem + 2 * kIN;                  // [2 buffers][36][4][16][2][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 3, hx = wave >> 2;     // cout block; row half of the xi this wave accumulates (= its role = the tile block it finishes)
    const int Wp = args()->a.Wp, Hp = args()->a.Hp, H = Hp - 2, W = Wp - 2, plane = Hp * Wp;
    const int T = args()->nrun;
    constexpr bool ABL = PROBE == 1, TR = PROBE >= 2, NOST = PROBE == 3;   // 1: ablation bits; 2: per-phase cycle sums of wave 0 (registers only, written once at the end); 3: 2 without output stores
    const int probe = ABL ? args()->probe : 0;
    unsigned long long sg_t = 0, sg_sum[4] = {0, 0, 0, 0};   // TR: cycles inside the chunk body: groups [0,2) [2,8) [8,10) [10,18)
    unsigned long long ph_t = 0, ph_sum[4] = {0, 0, 0, 0};   // TR: 0 chunk bodies, 1 waits at the chunk barrier, 2 epilogue, 3 tile advance
    auto phase = [&](int i) {
        if constexpr (TR) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (i >= 0) ph_sum[i] += now - ph_t;
            ph_t = now;
        }
    };

    const int P = gridDim.x, xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3;
    const int nx = (P >> 3) + (xcd < (P & 7) ? 1 : 0);
    const int ntiles = args()->ntiles, per = ntiles >> 3, rem = ntiles & 7;
    const int tcnt = per + (xcd < rem ? 1 : 0), tstart = xcd * per + (xcd < rem ? xcd : rem);
    if (wslot >= tcnt) return;

    struct TileS {
        const float* seg0[2]; const float* seg1[2];   // planes of the two sub-regions' windows
        const float2* uw;                        // this wave's A operands: [chunk][xi][64 lanes]
        int base[2];                             // y0 * Wp + x0
        int n[2], ry[2], rx[2], cq, valid1;
    };
    auto mdiv = [](int x, unsigned long long m) { return (int)(((unsigned long long)(unsigned)x * m) >> 40); };
    auto tile_of = [&](int lid) {
        TileS t;
        const KArgs ka = args();
        const int q = mdiv(lid, ka->m_cq);
        t.cq = lid - q * ka->ncq;
        const int set = mdiv(q, ka->m_pps), pr = q - set * ka->pps;
        const int S = ka->S, RR = ka->RR, RXn = ka->RXn;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int s = 2 * pr + k;
            if (k == 1) t.valid1 = s < S ? 1 : 0;
            s = s < S ? s : S - 1;
            const int nl = mdiv(s, ka->m_rr), r2 = s - nl * RR;
            t.ry[k] = mdiv(r2, ka->m_rx);
            t.rx[k] = r2 - t.ry[k] * RXn;
            t.n[k] = set * ka->a.n_per_set + nl;
            t.seg0[k] = ka->a.seg[0].base + (long)nl * ka->a.seg[0].stride_n + ka->a.seg[0].set_off[set];
            t.seg1[k] = ka->a.seg[1].C > 0 ? ka->a.seg[1].base + (long)nl * ka->a.seg[1].stride_n + ka->a.seg[1].set_off[set] : t.seg0[k];
            t.base[k] = (16 * t.ry[k]) * Wp + 16 * t.rx[k];
        }
        t.uw = reinterpret_cast<const float2*>(ka->U + (long)set * ka->u_set_stride) + ((long)(t.cq * 4 + cb) * ka->nchunk) * (36 * 64);
        return t;
    };

    f32x4 acc[18][2];                            // [6 al + b][tile block]: xi = 6 (3 hx + al) + b
    auto a_load = [&](const float2* uw, int c, int k) { return uw[((long)c * 36 + 18 * hx + k) * 64 + lane]; };
    auto drain_loads = [] { __builtin_amdgcn_s_waitcnt(0x0F70); };   // vmcnt(0), expcnt / lgkmcnt untouched

    // ROLE 0 (waves 0..3): input transform of stream position p + 1 while position p is multiplied.  ROLE 1 (waves 4..7): staging of
    // position p + 2 (LDS stores) and p + 3 (global loads).  Both: 72 MFMAs per chunk, the A-operand ring, the epilogue of their own
    // (cout block, sub-region).  Separate straight-line instantiations of the WHOLE walk.
    auto walk = [&](auto ROLEC) {
        constexpr int ROLE = decltype(ROLEC)::value;
        // ---- staging (both roles).  Wave w stages the channel pair (q, q + 4), q = w & 3, of sub-region w >> 2 (= ROLE): load (h, j) = channel
        // q + 4 h, pixel pairs 64 j + lane (j < 3) of that channel's 162.  Every load instruction has ONE window, channel and segment: its
        // base is a scalar pointer, the lane offset a register that only depends on the tile.  Six loads per wave and chunk: the vector-
        // memory path takes ~90 cycles per such instruction (7 rows of 72 bytes), and a wave that issues twelve of them in a row sits
        // ~4 k cycles at its queue -- spread over all eight waves nobody is the straggler at the chunk barrier.
        const int q = wave & 3;
        int loff[3], lrc[3];                     // row * Wp + col (global) / (row * kIP + col) * 2 (LDS floats) of pixel pair 64 j + lane
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int e = 64 * j + lane;
            e = e < kSE ? e : kSE - 1;
            const int row = e / (kIR / 2), col = 2 * (e - row * (kIR / 2));
            loff[j] = row * Wp + col;
            lrc[j] = (row * kIP + col) * 2;
        }
        const bool st_tail = lane < kSE - 128;   // lanes of the j = 2 loads that hold a pixel pair
        // the six pieces (sub-region sb, j) = 3 sb + j of a channel pair are dealt to the SIMD's two waves so that both reach the chunk barrier
        // together: the transform wave takes (0, 0) and the short (0, 2), the other wave the remaining four (measured: 6 / 6 left the
        // transform waves 1.5 k cycles per chunk behind, 0 / 12 the others 3.4 k)
#ifndef TTC_W4_SPLIT
#define TTC_W4_SPLIT 2
#endif
        constexpr int NPC = ROLE == 0 ? TTC_W4_SPLIT : 6 - TTC_W4_SPLIT;
        constexpr int PCS[5] = {ROLE == 0 ? 0 : (TTC_W4_SPLIT == 1 ? 1 : 1), ROLE == 0 ? 2 : (TTC_W4_SPLIT == 1 ? 2 : 3), TTC_W4_SPLIT == 1 ? 3 : 4, TTC_W4_SPLIT == 1 ? 4 : 5, 5};
        int spo_cur[NPC], spo_nxt[NPC];
        float2 g[2 * NPC];                       // [h 2][piece]
        // positions past the plane's end (the 18 x 18 image of an edge sub-region) only feed outputs that are never stored: any finite
        // in-plane value will do, so the offset is clamped into the plane.  PRECONDITION (include/ttc.h, "Non-finite inputs"): the planes hold
        // finite values.  The transforms mix a patch's 36 inputs into every V, so a NaN / Inf picked up by a clamped read reaches STORED
        // outputs of a partly valid patch where the direct kernel would not read it -- but every in-plane position of a padded plane (and
        // every plane of the real channel Cin - 1 the pad channels alias below) is an input of some valid output of the SAME window, and
        // the layer's GroupNorm statistics span the whole window: the direct kernel turns that window into NaN as well.  What differs is
        // only WHICH raw outputs are non-finite before the statistics are taken, never which windows are (tested:
        // tests/test_gpu_model.py::test_non_finite_window_stays_in_its_window)
        auto lane_offsets = [&](const TileS& t, int (&spo)[NPC]) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) spo[i] = min(t.base[PCS[i] / 3] + loff[PCS[i] % 3], plane - 2);
        };
        // piece j of stream chunk c of the tile t (straight-line: selects, never a branch around a load)
        auto stage_load_piece = [&](const float* s0, const float* s1, int sp, int c, int i) {
            const int Cin = args()->a.Cin, C0 = args()->a.seg[0].C;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int ci = c * kCK + q + 4 * h;
                ci = ci < Cin ? ci : Cin - 1;    // pad channels meet zero weights: any finite plane will do
                const bool lo = ci < C0;
                const float* bp = (lo ? s0 : s1) + (long)(lo ? ci : ci - C0) * plane;     // scalar
                g[h * NPC + i] = *reinterpret_cast<const float2*>(bp + sp);
            }
        };
        auto stage_load = [&](const TileS& t, const int (&spo)[NPC], int c) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) stage_load_piece(t.seg0[PCS[i] / 3], t.seg1[PCS[i] / 3], spo[i], c, i);
        };
        // pixel pairs j of channel pair q: elements [x][s] and [x + 1][s] are four consecutive floats
        auto stage_store_piece = [&](int buf, int i) {
            const int sb = PCS[i] / 3, j = PCS[i] % 3;
            float* d = in_tile + buf * kIN + sb * kSUB + q * kKQ + lrc[j];
            if (j < 2 || st_tail) { d[0] = g[i].x; d[1] = g[NPC + i].x; d[2] = g[i].y; d[3] = g[NPC + i].y; }
        };
        // ---- ROLE 0: input transform.  Wave w: row half w & 1 (rows 3 h .. 3 h + 2 of B^T d, from input rows h .. h + 4) of the channel
        // pairs kq = 2 (w >> 1) + (lane >> 5); lane & 31 = tile (lane & 15 = tx + 4 ty, bit 4 = sub-region)
        const int thalf = wave & 1, tkq = 2 * ((wave >> 1) & 1) + (lane >> 5), tt16 = lane & 15, ttb = (lane >> 4) & 1;
        const float* tsrc = in_tile + ttb * kSUB + tkq * kKQ + ((4 * (tt16 >> 2) + thalf) * kIP + 4 * (tt16 & 3)) * 2;
        float* tdst = Vb + (18 * thalf) * 256 + tkq * 64 + tt16 * 4 + ttb * 2;
        v2f tt[3][6];                            // B^T d rows of this half
        // column pair p of the half patch -> tt[.][2 p], tt[.][2 p + 1].  The five 16-byte reads are issued two groups before the arithmetic
        // that consumes them (tr_load / tr_cols): issued together, every column pair exposed one LDS round trip to the wave
        v2f td[5][2];
        auto tr_load = [&](const float* tin, int p2) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float4 v = *reinterpret_cast<const float4*>(tin + (r * kIP + 2 * p2) * 2);
                td[r][0] = v2f{v.x, v.y}; td[r][1] = v2f{v.z, v.w};
            }
        };
        auto tr_cols = [&](int p2) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * p2 + cc;
                const v2f e0 = td[0][cc], e1 = td[1][cc], e2 = td[2][cc], e3 = td[3][cc], e4 = td[4][cc];
                if (thalf == 0) {                // rows 0, 1, 2 of B^T from input rows 0 .. 4
                    tt[0][c] = pk_fma(4.f, e0, pk_fma(-5.f, e2, e4));
                    tt[1][c] = pk_fma(-4.f, e1 + e2, e3 + e4);
                    tt[2][c] = pk_fma(4.f, e1 - e2, e4 - e3);
                } else {                         // rows 3, 4, 5 from input rows 1 .. 5 (e_r = input row r + 1)
                    const v2f m31 = e2 - e0, m42 = e3 - e1;
                    tt[0][c] = pk_fma(2.f, m31, m42);
                    tt[1][c] = pk_fma(-2.f, m31, m42);
                    tt[2][c] = pk_fma(4.f, e0, pk_fma(-5.f, e2, e4));
                }
            }
        };
        // row al of the half: (B^T d) B -> V[xi = 6 (3 h + al) + b], written as the float2 of the channel pair = the two k-steps
        auto tr_row = [&](float* tout, int al) {
            v2f o[6];
            bt6(tt[al], o);
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) *reinterpret_cast<v2f*>(tout + (6 * al + b2) * 256) = o[b2];
        };

        float2 A[kAR];                           // A-operand ring: slot k % kAR is refilled right after its four MFMAs with the operand kAR groups ahead
        int tk = wslot;
        TileS cur = tile_of(tstart + tk);
        bool has_next = tk + nx < tcnt;
        TileS nxt = tile_of(tstart + (has_next ? tk + nx : tk));
        lane_offsets(cur, spo_cur);
        lane_offsets(nxt, spo_nxt);

        // ---- start-up of the workgroup's first tile (T >= 3: checked at launch).  Invariant at the start of the chunk with parity par:
        // V[par] holds its stream position p, IN[par ^ 1] position p + 1, g[] position p + 2 (requested)
#pragma unroll
        for (int i = 0; i < kAR; ++i) A[i] = a_load(cur.uw, 0, i);
        stage_load(cur, spo_cur, 0);
#pragma unroll
        for (int i = 0; i < NPC; ++i) stage_store_piece(0, i);
        stage_load(cur, spo_cur, 1);
        __syncthreads();
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int p2 = 0; p2 < 3; ++p2) { tr_load(tsrc, p2); tr_cols(p2); }
#pragma unroll
            for (int al = 0; al < 3; ++al) tr_row(tdst, al);
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i) stage_store_piece(1, i);
        stage_load(cur, spo_cur, 2);

        const float4* Vr = reinterpret_cast<const float4*>(Vb) + 18 * hx * 64 + lane;
        int par = 0;

        // One chunk c of the running tile = 36 xi GROUPS of two MFMAs (the two k-steps of one A / B operand pair), one barrier at its start.
        // The role's other work is cut into pieces between the groups (sched_barrier fences: left alone the scheduler runs the whole
        // transform before the chunk's first MFMA and sinks the loads):
        //   every group: the B operand of group + 2 (ds_read_b64), the A operand of group + kRING (global, scalar base);
        //   ROLE 0, groups 1 / 5 / 9: B^T d of one column pair (5 ds_read_b128 + 14 packed ops); 14 / 20 / 26: one row of (B^T d) B
        //           (14 packed ops, 6 ds_write_b64 into V[par ^ 1]);
        //   ROLE 1, groups 6 .. 11: the staging stores of stream position c + 2 into IN[par]; group 30: the 12 loads of position c + 3.
        auto chunk = [&](int c, auto LASTC, auto FIRSTC) {
            constexpr bool last = decltype(LASTC)::value;
            constexpr bool first = decltype(FIRSTC)::value;
            phase(0);
            if (!ABL || !(probe & 64)) lds_barrier();   // V[par] and IN[par ^ 1] are complete; V[par ^ 1] and IN[par] are free
            phase(1);
            const float4* Vc = Vr + par * (kVB / 4);
            const float* tin = tsrc + (par ^ 1) * kIN;
            float* tout = tdst + (par ^ 1) * kVB;
            const bool in1 = !last, in3 = c + 3 < T;
            const int nks = last ? args()->nks_last : 2;
            const float2* uwn = in1 ? cur.uw : nxt.uw;
            const int cn = in1 ? c + 1 : 0;
            const bool do_tr = !ABL || !(probe & 2);
            float4 b[3];
            b[0] = Vc[0]; b[1] = Vc[64];
            if constexpr (TR) sg_t = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                if constexpr (TR) { if (k == 2 || k == 8 || k == 10) { const unsigned long long now = __builtin_amdgcn_s_memtime(); sg_sum[k == 2 ? 0 : (k == 8 ? 1 : 2)] += now - sg_t; sg_t = now; } }
                if (k + 2 < 18 && (!ABL || !(probe & 32))) b[(k + 2) % 3] = Vc[(k + 2) * 64];
                const float2 a = A[k % kAR];
                const float4 bb = b[k % 3];
                if (first) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, z, 0, 0, 0);
                    acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.z, z, 0, 0, 0);
                } else {
                    acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, acc[k][0], 0, 0, 0);
                    acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.z, acc[k][1], 0, 0, 0);
                }
                if (!last || nks > 1) {
                    acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, acc[k][0], 0, 0, 0);
                    acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.w, acc[k][1], 0, 0, 0);
                }
#ifndef TTC_W4_NOA
                if (!ABL || !(probe & 16)) A[k % kAR] = k + kAR < 18 ? a_load(cur.uw, c, k + kAR) : a_load(uwn, cn, k + kAR - 18);
#endif
                // ---- this group's piece of the role's other work
                if constexpr (ROLE == 0) {
                    if (do_tr) {
                        if (k == 0) tr_load(tin, 0);
                        if (k == 2 || k == 4 || k == 6) { tr_cols(k / 2 - 1); if (k < 6) tr_load(tin, k / 2); }
                        if (k == 8 || k == 11 || k == 14) tr_row(tout, (k - 8) / 3);
                    }
                }
                // staging piece j: its two registers are stored (stream position c + 2, requested a chunk ago) and re-requested at once for
                // position c + 3 (past the tile's end that is chunk c + 3 - T of the next tile)
                constexpr int K0 = ROLE == 0 ? 9 : 2, KS = ROLE == 0 ? 4 : (NPC == 5 ? 2 : 3);   // groups that carry a staging piece: K0, K0 + KS, ...
                if (k >= K0 && (k - K0) % KS == 0 && (k - K0) / KS < NPC) {
                    const int i = (k - K0) / KS;
                    const int sb = PCS[i] / 3;
                    stage_store_piece(par, i);
#ifndef TTC_W4_NOSTAGE
                    if (!ABL || !(probe & 8))
                        stage_load_piece(in3 ? cur.seg0[sb] : nxt.seg0[sb], in3 ? cur.seg1[sb] : nxt.seg1[sb], in3 ? spo_cur[i] : spo_nxt[i],
                                         in3 ? c + 3 : c + 3 - T, i);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (TR) { const unsigned long long now = __builtin_amdgcn_s_memtime(); sg_sum[3] += now - sg_t; }
            par ^= 1;
        };

        unsigned long long tr0 = 0, tr1 = 0; int ntl = 0;
        if constexpr (TR) { tr0 = __builtin_amdgcn_s_memrealtime(); tr1 = __builtin_amdgcn_s_memtime(); }
        phase(-1);
        for (;;) {
            chunk(0, std::false_type{}, std::true_type{});
            for (int c = 1; c + 1 < T; ++c) chunk(c, std::false_type{}, std::false_type{});
            chunk(T - 1, std::true_type{}, std::false_type{});

            // ---- epilogue.  This wave holds M[a][b] for its three rows a of BOTH tile blocks.  Row half first (local):
            //   Z[a][j] = sum_b M[a][b] A[b][j]; the Z of tile block 1 - hx goes to the partner wave (same SIMD, other xi half) through LDS,
            //   then Y[i][j] = sum_a A^T[i][a] Z[a][j] over all six a for the own tile block hx, epilogue op, GroupNorm sums, stores.
            // Exchange area = what is idle between two chunks: V[par ^ 1] (the last chunk's operand buffer), IN[par] (read by the last
            // chunk's transform, rewritten only behind the next chunk's barrier) and the 7 half-slots kept behind the buffers: 16 half-slots
            // of 6 KB (8 waves x 2 cout pairs), ALL written before ONE barrier and read behind it.
            phase(0);
            float4 zk[4][3];                         // own tile block: Z[r][al]
            if (!ABL || !(probe & 4)) {
                if (!ABL || !(probe & 64)) lds_barrier();   // every wave has finished the last chunk's reads of V[par ^ 1] / IN[par]
                auto slot = [&](int kx) -> float4* {
                    float* base = kx < 6 ? Vb + (par ^ 1) * kVB + kx * kXS
                                         : (kx < 13 ? Vb + 2 * kVB + (kx - 6) * kXS : in_tile + par * kIN + (kx - 13) * kXS);
                    return reinterpret_cast<float4*>(base) + lane;
                };
                auto zrow = [](float m0, float m1, float m2, float m3, float m4, float m5) {
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    return float4{m0 + s12 + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + m5};
                };
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    float4* sw = slot(2 * wave + q2);
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int r = 2 * q2 + rr;
#pragma unroll
                        for (int al = 0; al < 3; ++al) {
                            sw[(rr * 3 + al) * 64] = zrow(acc[6 * al][1 - ROLE][r], acc[6 * al + 1][1 - ROLE][r], acc[6 * al + 2][1 - ROLE][r],
                                                          acc[6 * al + 3][1 - ROLE][r], acc[6 * al + 4][1 - ROLE][r], acc[6 * al + 5][1 - ROLE][r]);
                            zk[r][al] = zrow(acc[6 * al][ROLE][r], acc[6 * al + 1][ROLE][r], acc[6 * al + 2][ROLE][r],
                                             acc[6 * al + 3][ROLE][r], acc[6 * al + 4][ROLE][r], acc[6 * al + 5][ROLE][r]);
                        }
                    }
                }
                if (!ABL || !(probe & 64)) lds_barrier();
            }
            if ((!ABL || !(probe & 4)) && (ROLE == 0 || cur.valid1)) {
                const KArgs ka = args();
                float* const out = ka->a.out;
                const long out_stride_n = ka->a.out_stride_n, out_plane = ka->a.out_plane;
                float* const stats = ka->a.stats;
                const int Cout = ka->a.Cout, same_pad = ka->a.same_pad, RXn = ka->RXn, RR = ka->RR;
                const int t16 = lane & 15, ty = t16 >> 2, tx = t16 & 3, qd = lane >> 4;
                const int sry = ROLE ? cur.ry[1] : cur.ry[0], srx = ROLE ? cur.rx[1] : cur.rx[0], sn = ROLE ? cur.n[1] : cur.n[0];
                const int pw = (wave + 4) & 7;       // the partner
                auto slot_r = [&](int kx) -> const float4* {
                    const float* base = kx < 6 ? Vb + (par ^ 1) * kVB + kx * kXS
                                               : (kx < 13 ? Vb + 2 * kVB + (kx - 6) * kXS : in_tile + par * kIN + (kx - 13) * kXS);
                    return reinterpret_cast<const float4*>(base) + lane;
                };
                typedef float pair_f __attribute__((ext_vector_type(2), aligned(8)));
                // WHOLE: the sub-region lies inside the image (wave-uniform) -- no per-store predicates, no per-sum selects.  The two
                // forms are separate instantiations so that the compiler cannot merge them back into predicated single-store blocks.
                auto sub_epilogue = [&](auto WHOLEC) {
                    constexpr bool whole = decltype(WHOLEC)::value;
                    const int y0 = 16 * sry + 4 * ty, x0 = 16 * srx + 4 * tx;
                    // a tile's columns are valid in pairs (W is even: checked at launch): (x0, x0 + 1) and (x0 + 2, x0 + 3)
                    const bool vc0 = x0 < W, vc1 = x0 + 2 < W;
                    float* const ob = out + (long)sn * out_stride_n + (long)((cur.cq * 4 + cb) * 16 + 4 * qd) * out_plane + (long)y0 * Wp + x0;
                    float ssum = 0.f, ssq = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float4* sr = slot_r(2 * pw + (r >> 1));
                        float4 z[6];                 // Z[a][.] for a = 0 .. 5: own rows 3 hx + al, the partner's 3 (1 - hx) + al
#pragma unroll
                        for (int al = 0; al < 3; ++al) {
                            z[3 * ROLE + al] = zk[r][al];
                            z[3 * (1 - ROLE) + al] = sr[((r & 1) * 3 + al) * 64];
                        }
                        float o4[4][4];              // Y[i][j]
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float z0 = z[0][j], z1 = z[1][j], z2 = z[2][j], z3 = z[3][j], z4 = z[4][j], z5 = z[5][j];
                            const float s12 = z1 + z2, d12 = z1 - z2, s34 = z3 + z4, d34 = z3 - z4;
                            o4[0][j] = z0 + s12 + s34;
                            o4[1][j] = fmaf(2.f, d34, d12);
                            o4[2][j] = fmaf(4.f, s34, s12);
                            o4[3][j] = fmaf(8.f, d34, d12) + z5;
                        }
                        float* const oc = ob + (long)r * out_plane;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float o[4] = {o4[i][0], o4[i][1], o4[i][2], o4[i][3]};
                            if (EPI == EPI_SWISH) {
                                if (same_pad) {
                                    const int y = y0 + i;
                                    const bool ey = (y == 0) || (y == H - 1);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const bool ex = (x0 + j == 0) || (x0 + j == W - 1);
                                        o[j] *= (ey && ex) ? 2.25f : ((ey || ex) ? 1.5f : 1.0f);
                                    }
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = o[j] * sigmoidf_(o[j]);
                            }
                            const float s01 = o[0] + o[1], s23 = o[2] + o[3];
                            const float q01 = fmaf(o[1], o[1], o[0] * o[0]), q23 = fmaf(o[3], o[3], o[2] * o[2]);
                            pair_f p0, p1; p0.x = o[0]; p0.y = o[1]; p1.x = o[2]; p1.y = o[3];
                            if (whole) {
                                ssum += s01 + s23; ssq += q01 + q23;
                                if (!NOST && (!ABL || !(probe & 1))) {
                                    *reinterpret_cast<pair_f*>(oc + (long)i * Wp) = p0;
                                    *reinterpret_cast<pair_f*>(oc + (long)i * Wp + 2) = p1;
                                }
                            } else {
                                const bool vr = y0 + i < H;
                                const bool ok0 = vr && vc0, ok1 = vr && vc1;
                                ssum += (ok0 ? s01 : 0.f) + (ok1 ? s23 : 0.f);
                                ssq += (ok0 ? q01 : 0.f) + (ok1 ? q23 : 0.f);
                                if (!NOST && (!ABL || !(probe & 1))) {
                                    if (ok0) *reinterpret_cast<pair_f*>(oc + (long)i * Wp) = p0;
                                    if (ok1) *reinterpret_cast<pair_f*>(oc + (long)i * Wp + 2) = p1;
                                }
                            }
                        }
                    }
                    if (stats) {                     // a lane's four couts are one GroupNorm quad: sum over the 16 lanes (tiles) of its DPP row
                        float red[2] = {ssum, ssq};
#pragma unroll
                        for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x111, 0xf>(red[i]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x112, 0xf>(red[i]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x114, 0xf>(red[i]);
#pragma unroll
                        for (int i = 0; i < 2; ++i) red[i] += dpp_shift<0x118, 0xf>(red[i]);
                        if (t16 == 15) {
                            const int quad = (cur.cq * 4 + cb) * 4 + qd;
                            float2* dst = reinterpret_cast<float2*>(stats) + ((long)sn * (Cout / 4) + quad) * RR + sry * RXn + srx;
                            *dst = make_float2(red[0], red[1]);
                        }
                    }
                };
                const bool whole = (16 * sry + 16 <= H) && (16 * srx + 16 <= W);
                if (whole) sub_epilogue(std::true_type{}); else sub_epilogue(std::false_type{});
            }
            phase(2);
            ++ntl;
            if (!has_next) break;
            tk += nx;
            cur = nxt;
#pragma unroll
            for (int i = 0; i < NPC; ++i) spo_cur[i] = spo_nxt[i];
            has_next = tk + nx < tcnt;
            if (has_next) { nxt = tile_of(tstart + tk + nx); lane_offsets(nxt, spo_nxt); }
            phase(3);
        }
        if constexpr (TR) {
            unsigned long long* tr = args()->trace;
            if (tr && lane == 0 && (wave & 3) == 0) {
                tr += (long)blockIdx.x * 32 + (wave >> 2) * 16;
                tr[8] = sg_sum[0]; tr[9] = sg_sum[1]; tr[10] = sg_sum[2]; tr[11] = sg_sum[3];
                tr[0] = ph_sum[0]; tr[1] = ph_sum[1]; tr[2] = ph_sum[2]; tr[3] = ph_sum[3]; tr[4] = (unsigned long long)ntl;
                tr[5] = __builtin_amdgcn_s_memrealtime() - tr0; tr[6] = __builtin_amdgcn_s_memtime() - tr1; tr[7] = (unsigned long long)T;
            }
        }
    };
    if (wave < 4) walk(std::integral_constant<int, 0>{});
    else walk(std::integral_constant<int, 1>{});
}

int cus_of_device() {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    static int cache[64] = {0};
    if (dev >= 0 && dev < 64 && cache[dev]) return cache[dev];
    (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    if (dev >= 0 && dev < 64) cache[dev] = v;
    return v;
}

template <int EPI>
hipError_t launch_w4(const ConvArgs& a, const PackedConv& pw, int n, hipStream_t s) {
    const size_t lds = (size_t)(2 * kIN + 2 * kVB + 7 * kXS) * sizeof(float);
    static LdsConfig lds_cfg, lds_cfg_p;
    const int H = a.Hp - 2, W = a.Wp - 2;
    const int RXn = (W + 15) / 16, RYn = (H + 15) / 16, ncq = a.Cout / 64;
    const int cin_run = (a.cin_run > 0 && a.cin_run < a.Cin) ? a.cin_run : a.Cin;
    const int nrun = std::max(3, (cin_run + kCK - 1) / kCK);
    const int rem = std::min(a.Cin, nrun * kCK) - kCK * (nrun - 1);
    const int nks_last = (rem + 3) / 4;
    const int nsets = (n + a.n_per_set - 1) / a.n_per_set;
    const long RR = (long)RXn * RYn, S = RR * std::min(n, a.n_per_set), pps = (S + 1) / 2;
    const long ntiles = pps * nsets * ncq;
    auto magic = [](long d) { return (1ULL << 40) / (unsigned long long)d + 1ULL; };
    Wino4Args wa{a, pw.d_wu4, pw.set_stride_w4, pw.nchunk_w4, nks_last, RXn, RYn, ncq, (int)ntiles, (int)RR, (int)S, (int)pps,
                 magic(ncq), magic(pps), magic(RR), magic(RXn), nrun, cin_run, nullptr, 0};
    static const int probe = [] { const char* e = getenv("TTC_WINO4_PROBE"); return e ? atoi(e) : 0; }();
    static const int grid_force = [] { const char* e = getenv("TTC_WINO4_GRID"); return e ? atoi(e) : 0; }();
    const long grid = std::min<long>(ntiles, grid_force > 0 ? grid_force : cus_of_device());
    {   // TTC_WINO4_TRACE=1: per-phase cycle sums of every workgroup's wave 0 for the launches of epilogue kind TTC_WINO4_TRACE_EPI (default 0 = gates),
        // printed as one line per traced launch (median over workgroups); the first TTC_WINO4_TRACE_SKIP matching launches pass untraced
        static const int tr_on = [] { const char* e = getenv("TTC_WINO4_TRACE"); return e ? atoi(e) : 0; }();   // 1: trace; 2: trace without output stores
        static const int tr_epi = [] { const char* e = getenv("TTC_WINO4_TRACE_EPI"); return e ? atoi(e) : (int)EPI_RAW; }();
        static const int tr_cin = [] { const char* e = getenv("TTC_WINO4_TRACE_CIN"); return e ? atoi(e) : 0; }();
        static int tr_skip = [] { const char* e = getenv("TTC_WINO4_TRACE_SKIP"); return e ? atoi(e) : 8; }();
        static int tr_left = [] { const char* e = getenv("TTC_WINO4_TRACE_N"); return e ? atoi(e) : 2; }();
        const bool match = tr_on && EPI == tr_epi && (tr_cin == 0 || tr_cin == a.Cin) && tr_left > 0;
        if (match && tr_skip > 0) tr_skip--;
        else if (match) {
            tr_left--;
            static LdsConfig lds_cfg_t, lds_cfg_t3;
            unsigned long long* d = nullptr;
            const size_t bytes = (size_t)grid * 32 * sizeof(unsigned long long);
            (void)hipStreamSynchronize(s);
            if (lds_cfg_t.ensure(&conv3x3_wino4<EPI, 2>, lds) == hipSuccess && lds_cfg_t3.ensure(&conv3x3_wino4<EPI, 3>, lds) == hipSuccess && hipMalloc(&d, bytes) == hipSuccess) {
                (void)hipMemset(d, 0, bytes);
                wa.trace = d;
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                float ms = 0.f;
                (void)hipEventRecord(e0, s);
                if (tr_on == 2) hipLaunchKernelGGL((conv3x3_wino4<EPI, 3>), dim3((unsigned)grid), dim3(kThreads4), lds, s, wa);
                else hipLaunchKernelGGL((conv3x3_wino4<EPI, 2>), dim3((unsigned)grid), dim3(kThreads4), lds, s, wa);
                (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s); (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h((size_t)grid * 32);
                (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost); (void)hipFree(d);
                for (int role = 0; role < 2; ++role) {
                    auto med = [&](int k) { std::vector<unsigned long long> v; for (long i = 0; i < grid; ++i) v.push_back(h[i * 32 + role * 16 + k]); std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; };
                    auto mx = [&](int k) { unsigned long long m = 0; for (long i = 0; i < grid; ++i) m = std::max(m, h[i * 32 + role * 16 + k]); return (double)m; };
                    const double tl = med(4), ghz = med(6) / (med(5) * 10.0);   // s_memrealtime ticks at 100 MHz
                    fprintf(stderr, "[wino4 trace] epi %d Cin %d Cout %d %dx%d n %d wave %d: %.3f ms, grid %ld, tiles/wg %.0f (max %.0f), chunks/tile %d, clock %.2f GHz | per tile (cycles, median wg): "
                                    "chunk bodies %.0f (%.0f per chunk: groups 0-1 %.0f, 2-7 %.0f, 8-9 %.0f, 10-17 %.0f), barrier waits %.0f, epilogue %.0f, advance %.0f | walk total %.0f (max %.0f)\n",
                            EPI, a.Cin, a.Cout, a.Hp - 2, a.Wp - 2, n, role * 4, ms, grid, tl, mx(4), nrun, ghz, med(0) / tl, med(0) / tl / nrun, med(8) / tl / nrun, med(9) / tl / nrun, med(10) / tl / nrun, med(11) / tl / nrun, med(1) / tl, med(2) / tl, med(3) / tl, med(6), mx(6));
                }
                wa.trace = nullptr;
                return hipGetLastError();
            }
        }
    }
    if (probe) {
        wa.probe = probe;
        if (hipError_t e = lds_cfg_p.ensure(&conv3x3_wino4<EPI, 1>, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL((conv3x3_wino4<EPI, 1>), dim3((unsigned)grid), dim3(kThreads4), lds, s, wa);
        return hipGetLastError();
    }
    if (hipError_t e = lds_cfg.ensure(&conv3x3_wino4<EPI, 0>, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((conv3x3_wino4<EPI, 0>), dim3((unsigned)grid), dim3(kThreads4), lds, s, wa);
    return hipGetLastError();
}

}  // namespace

// the launch limits of the F(4x4) kernels: one predicate shared by conv_launch, conv_stat_slots_for and the ConvGRU step-0 shortcut
bool conv_wino4_ok(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set) {
    if (!pw.d_wu4 || pw.nchunk_w4 < 3 || (epi != EPI_RAW && epi != EPI_SWISH) || pw.Cout % 64 != 0) return false;
    if ((Wp & 1) || Wp < 6 || Hp < 6 || Wp >= (1 << 14)) return false;
    if ((long)Hp * Wp >= (1L << 24) || (long)Cin * Hp * Wp >= (1L << 31)) return false;
    if (n < 1 || n_per_set < 1 || n_per_set >= 4096) return false;
    const long RR = (long)((Wp - 2 + 15) / 16) * ((Hp - 2 + 15) / 16);
    const long S = RR * std::min(n, n_per_set), nsets = (n + n_per_set - 1) / n_per_set;
    // x / d == (x * magic(d)) >> 40 needs x * d < 2^40: ids stay below 2^24, divisors below 2^16
    if (RR >= (1L << 16) || (S + 1) / 2 >= (1L << 16) || nsets > 256 || ((S + 1) / 2) * nsets * (pw.Cout / 64) >= (1L << 24)) return false;
    if (n > n_per_set && n % n_per_set != 0) return false;     // every weight set holds the same number of windows
    return true;
}

// GroupNorm partial-sum slots per (window, channel quad) of the F(4x4) kernels: one per 16 x 16-pixel sub-region
int conv_wino4_stat_slots(int Hp, int Wp) { return ((Wp - 2 + 15) / 16) * ((Hp - 2 + 15) / 16); }

// U = G g G^T per (cin, cout) in double, packed [set][cout block 16][chunk][xi 36][kq 4][cout 16][s 2]: channel 8 c + 4 s + kq
long conv_pack_wino4(const float* const* hwio, int nsets, int Cin, int Cout, std::vector<float>& out, int* nchunk_out) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nchunk = (Cin + kCK - 1) / kCK, ncb = (Cout + 15) / 16;
    const long per_set = (long)ncb * nchunk * 36 * 128;
    out.assign((size_t)per_set * nsets, 0.0f);
    for (int s = 0; s < nsets; ++s)
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci) {
                double g[3][3], t[6][3];
                for (int u = 0; u < 3; ++u)
                    for (int v = 0; v < 3; ++v) g[u][v] = hwio[s][((long)(u * 3 + v) * Cin + ci) * Cout + co];
                for (int aa = 0; aa < 6; ++aa)
                    for (int v = 0; v < 3; ++v) t[aa][v] = G[aa][0] * g[0][v] + G[aa][1] * g[1][v] + G[aa][2] * g[2][v];
                const int cb = co >> 4, col = co & 15, c = ci / kCK, k = ci % kCK, kq = k & 3, st = k >> 2;
                for (int aa = 0; aa < 6; ++aa)
                    for (int bb = 0; bb < 6; ++bb) {
                        const double u = t[aa][0] * G[bb][0] + t[aa][1] * G[bb][1] + t[aa][2] * G[bb][2];
                        const int xi = aa * 6 + bb;
                        out[(size_t)s * per_set + (((((long)cb * nchunk + c) * 36 + xi) * 4 + kq) * 16 + col) * 2 + st] = (float)u;
                    }
            }
    if (nchunk_out) *nchunk_out = nchunk;
    return per_set;
}

hipError_t conv_launch_wino4(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s) {
    if (!conv_wino4_ok(pw, epi, a.Hp, a.Wp, a.Cin, n, a.n_per_set)) return hipErrorInvalidValue;
    if (epi == EPI_RAW) return launch_w4<EPI_RAW>(a, pw, n, s);
    return launch_w4<EPI_SWISH>(a, pw, n, s);
}
