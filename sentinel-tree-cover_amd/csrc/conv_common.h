// Shared pieces of the two conv engines (conv3x3_mfma.hip: exact fp32 MFMA; conv3x3_h16.hip: fp16 / bf16 pairs):
// tile geometry, XCD-aware tile order and the fused epilogues.  Both engines hold the same accumulator layout
// (v_mfma 32x32 D tile: row = cout (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column = pixel lane & 31).
#pragma once
#include "ttc_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace ttcconv {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kQG = 4;                       // pixel groups (of 32) per wave
constexpr int kBQ = kWaves * kQG * 32;       // 512 flattened positions per workgroup

// hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each, absolute error of the sigmoid <= 2e-7): libm's expf and the IEEE
// division cost ~40 VALU instructions and enough temporaries to spill the 128 accumulators of the swish epilogue
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// XCD-aware tile order: the dispatcher deals workgroup ids round-robin over the 8 XCDs, so id -> (id % 8) owns
// the contiguous slice [start(xcd), ...) of the logical tile list.  Neighbouring q tiles (which share a
// 2*Wp+2 halo, +68 % input bytes at Wp = 174) then meet in ONE XCD's L2 instead of being fetched from HBM twice.
__device__ __forceinline__ void tile_index(int nblk_q, int ncb, int& bq, int& cb, int& n) {
    const int G = gridDim.x, id = blockIdx.x;
    const int per = G >> 3, rem = G & 7, xcd = id & 7, slot = id >> 3;
    const int lid = xcd * per + (xcd < rem ? xcd : rem) + slot;
    bq = lid % nblk_q;
    const int rest = lid / nblk_q;
    cb = rest % ncb; n = rest / ncb;
}

// Sum over the 32 lanes of each half-wave with DPP row shifts (VALU, no LDS round trips): afterwards lane 31 holds the sum of
// lanes 0..31 and lane 63 the sum of lanes 32..63, for every one of the N values (the N chains interleave, so the DPP wait
// states are filled).  The ds_bpermute butterfly this replaces (5 dependent LDS round trips per value, 80 per tile) took
// 12 k cycles of a 90 k-cycle tile of the 16-bit engine.  Fixed association order: bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int N>
__device__ __forceinline__ void half_wave_sums(float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shift<0x111, 0xf>(v[i]);          // row_shr:1  (rows of 16 lanes, out-of-row sources read 0)
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shift<0x112, 0xf>(v[i]);          // row_shr:2
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shift<0x114, 0xf>(v[i]);          // row_shr:4
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shift<0x118, 0xf>(v[i]);          // row_shr:8  -> lane 15 of each row = row total
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shift<0x142, 0xa>(v[i]);          // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63
}

// fused epilogue: EPI op, output store into the consumer's (padded) plane, deterministic GroupNorm partial sums.
template <int NCG, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[NCG][kQG], int n, int cb, int bq,
                                              int nblk_q, const float* aux, int tid) {
    constexpr int BN = NCG * 32;
    const int Wp = a.Wp, Hp = a.Hp;
    // tid: thread index within the 4-wave group that owns the tile
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int q0 = bq * kBQ;
    const int Hout = Hp - 2, Wout = Wp - 2;
    float ssum[NCG][4], ssq[NCG][4];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) { ssum[g][k] = 0.f; ssq[g][k] = 0.f; }

    float* outn = a.out + (long)n * a.out_stride_n;
    const float* resn = (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_TANH_ADD) ? a.res + (long)n * a.out_stride_n : nullptr;
    // bias of this lane's 16 output channels per cout group, fetched ONCE and unconditionally (a load under `if (co < Cout)`
    // inside the pixel loop cost a memory round trip each: 25 k of a 40 k-cycle tile in the 16-bit DSen2 layers)
    float bias[NCG][16];
    if (EPI >= EPI_BIAS) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float bv = aux[co < a.Cout ? co : 0];
                bias[g][r] = co < a.Cout ? bv : 0.f;
            }
    }

#pragma unroll
    for (int j = 0; j < kQG; ++j) {
        const int q = q0 + (wave * kQG + j) * 32 + lo;
        const int y = q / Wp, x = q - y * Wp;
        const bool valid = (x < Wout) && (y < Hout);
        float gate = 1.0f;
        if (EPI == EPI_SSE) {
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += aux[(r & 3) + 8 * (r >> 2) + 4 * hi] * acc[0][j][r];
            dot += __shfl_xor(dot, 32);
            gate = sigmoidf_(dot);
        }
        float ratio = 1.0f;
        if (EPI == EPI_SWISH && a.same_pad) {
            const bool ey = (y == 0) || (y == Hout - 1), ex = (x == 0) || (x == Wout - 1);
            ratio = (ey && ex) ? 2.25f : ((ey || ex) ? 1.5f : 1.0f);
        }
        const long opix = (long)(y + a.oy) * a.out_pitch + (x + a.ox);
        // a.reflect_out: the output plane carries a 1-px REFLECT rim (MirrorPad before the next VALID conv); the pixels one
        // step inside the edge also write their mirror images, which replaces a separate rim-fill pass over every plane
        long dup_y = 0, dup_x = 0;
        bool any_dup = false;
        if (EPI >= EPI_BIAS && a.reflect_out) {
            if (valid) {
                dup_y = (y == 1) ? -2L * a.out_pitch : ((y == Hout - 2) ? 2L * a.out_pitch : 0L);
                dup_x = (x == 1) ? -2L : ((x == Wout - 2) ? 2L : 0L);
            }
            any_dup = __any((dup_y != 0) || (dup_x != 0));
        }
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            float rv[16];
            if (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_TANH_ADD) {
                // all 16 residual loads in flight at once, branch-free (a load under `if (ok)` costs a full
                // memory round trip each: BIAS_RES was 43 % slower than BIAS_RELU on the same conv)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = valid && (co < a.Cout);
                    rv[r] = resn[ok ? (long)co * a.out_plane + opix : 0];
                }
            }
            float vv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float v = acc[g][j][r];
                if (EPI == EPI_SSE) v *= gate;
                if (EPI == EPI_SWISH) { v *= ratio; v = v * sigmoidf_(v); }
                if (EPI >= EPI_BIAS) {
                    v += bias[g][r];
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    if (EPI == EPI_BIAS_RES) v = rv[r] + 0.1f * v;
                    if (EPI == EPI_BIAS_TANH_ADD) v = rv[r] + tanhf(v);
                }
                vv[r] = v;
                if (EPI <= EPI_SWISH && valid) { ssum[g][r >> 2] += v; ssq[g][r >> 2] += v * v; }
            }
            // stores: ONE exec-masked region per accumulator tile when the whole cout group is real (every layer but
            // DSen2's 6-channel head), instead of a branch around each of the 16 stores
            const bool group_full = (cb * BN + g * 32 + 32 <= a.Cout);
            if (group_full) {
                if (valid) {
                    float* o = outn + (long)(cb * BN + g * 32 + 4 * hi) * a.out_plane + opix;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[(long)((r & 3) + 8 * (r >> 2)) * a.out_plane] = vv[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (valid && co < a.Cout) outn[(long)co * a.out_plane + opix] = vv[r];
                }
            }
            if (EPI >= EPI_BIAS && any_dup) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cb * BN + g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (valid && co < a.Cout) {
                        float* o = outn + (long)co * a.out_plane + opix;
                        if (dup_y) o[dup_y] = vv[r];
                        if (dup_x) o[dup_x] = vv[r];
                        if (dup_y && dup_x) o[dup_y + dup_x] = vv[r];
                    }
                }
            }
        }
    }

    if (EPI <= EPI_SWISH && a.stats) {
        // reduce over the 32 pixel-lanes of each half-wave; one partial per WAVE (no LDS, no barrier: the epilogue of one
        // wave group may run while another group of the same workgroup is still in its MFMA phase).
        // stats: [n][Cout/4][nblk_q * kWaves][2], reduced in double by k_gn_finalize -> deterministic.
        float red[NCG * 8];
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(g * 4 + k) * 2] = ssum[g][k]; red[(g * 4 + k) * 2 + 1] = ssq[g][k]; }
        half_wave_sums(red);
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int quad = cb * (BN / 4) + g * 8 + 2 * k + hi;
                if (lo == 31 && quad * 4 < a.Cout) {
                    float* dst = a.stats + (((long)n * (a.Cout / 4) + quad) * (nblk_q * kWaves) + bq * kWaves + wave) * 2;
                    dst[0] = red[(g * 4 + k) * 2]; dst[1] = red[(g * 4 + k) * 2 + 1];
                }
            }
    }
}

// LDS bytes conv_epilogue_flat needs (one 32-cout group of a tile, rows padded by 4 floats)
constexpr int kFlatRow = kBQ + 4;
constexpr size_t kFlatLdsBytes = (size_t)32 * kFlatRow * sizeof(float);

// Epilogue of the GroupNorm layers when the output keeps the input pitch (ConvArgs.flat_out): the tile's 512 positions are 512
// CONSECUTIVE floats of every output plane.  The accumulator layout (lane = one position, 16 couts) would leave as 4-byte
// stores -- 128 store instructions per wave whose issue, not HBM, bounds the tile's tail (measured: 4.8 us of a 23 us tile).
// Instead each 32-cout group goes through LDS ([cout][position], free after the main loop) and leaves as rows:
// wave w stores couts 8w .. 8w+7, a lane 4 consecutive positions -> 16 global_store_dwordx4 of 1 KiB per wave and group.
// EPI op and the deterministic GroupNorm partial sums are the same as in conv_epilogue.
template <int NCG, int EPI>
__device__ __forceinline__ void conv_epilogue_flat(const ConvArgs& a, f32x16 (&acc)[NCG][kQG], int n, int cb, int bq, int nblk_q,
                                                   const float* aux, int tid, float* lds, unsigned long long* tq = nullptr) {
    static_assert(EPI <= EPI_SWISH, "flat output is for the GroupNorm layers");
    if (tq) tq[0] = __builtin_amdgcn_s_memtime();
    constexpr int BN = NCG * 32;
    const int Wp = a.Wp, Hp = a.Hp;
    const int lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
    const int q0 = bq * kBQ;
    const int Hout = Hp - 2, Wout = Wp - 2;
    float ssum[NCG][4], ssq[NCG][4];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) { ssum[g][k] = 0.f; ssq[g][k] = 0.f; }
#pragma unroll
    for (int j = 0; j < kQG; ++j) {
        const int q = q0 + (wave * kQG + j) * 32 + lo;
        const int y = q / Wp, x = q - y * Wp;
        const bool valid = (x < Wout) && (y < Hout);
        float gate = 1.0f;
        if (EPI == EPI_SSE) {
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += aux[(r & 3) + 8 * (r >> 2) + 4 * hi] * acc[0][j][r];
            dot += __shfl_xor(dot, 32);
            gate = sigmoidf_(dot);
        }
        float ratio = 1.0f;
        if (EPI == EPI_SWISH && a.same_pad) {
            const bool ey = (y == 0) || (y == Hout - 1), ex = (x == 0) || (x == Wout - 1);
            ratio = (ey && ex) ? 2.25f : ((ey || ex) ? 1.5f : 1.0f);
        }
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[g][j][r];
                if (EPI == EPI_SSE) v *= gate;
                if (EPI == EPI_SWISH) { v *= ratio; v = v * sigmoidf_(v); }
                acc[g][j][r] = v;
                if (valid) { ssum[g][r >> 2] += v; ssq[g][r >> 2] += v * v; }
            }
    }
    if (tq) tq[6] = __builtin_amdgcn_s_memtime();
    if (a.stats) {
        float red[NCG * 8];
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(g * 4 + k) * 2] = ssum[g][k]; red[(g * 4 + k) * 2 + 1] = ssq[g][k]; }
        half_wave_sums(red);
        if (tq) tq[7] = __builtin_amdgcn_s_memtime();
        if (lo == 31) {            // one exec region for the two writer lanes; (sum, sumsq) leave as one 8-byte store
            const long slots = (long)nblk_q * kWaves;
            float2* base = reinterpret_cast<float2*>(a.stats) + (long)n * (a.Cout / 4) * slots + bq * kWaves + wave;
#pragma unroll
            for (int g = 0; g < NCG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int quad = cb * (BN / 4) + g * 8 + 2 * k + hi;
                    if (quad * 4 < a.Cout) base[quad * slots] = make_float2(red[(g * 4 + k) * 2], red[(g * 4 + k) * 2 + 1]);
                }
        }
    }
    if (tq) tq[1] = __builtin_amdgcn_s_memtime();
    float* outn = a.out + (long)n * a.out_stride_n;
    const long qend = (long)Hout * Wp;                 // positions of the rows that exist
#pragma unroll
    for (int g = 0; g < NCG; ++g) {
        __syncthreads();                               // LDS is free: every wave is past its last MFMA read / previous group
        if (tq && g == 0) tq[2] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < kQG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds[((r & 3) + 8 * (r >> 2) + 4 * hi) * kFlatRow + (wave * kQG + j) * 32 + lo] = acc[g][j][r];
        if (tq && g == 0) tq[3] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (tq && g == 0) tq[4] = __builtin_amdgcn_s_memtime();
        // rows leave as 16-byte vectors; only the LAST tile of a plane can run past its end, so the per-element guards live in a
        // separate (workgroup-uniform) path -- as per-store compares they cost 96 exec-masked branches per tile
        static_assert(kFlatRow % 4 == 0, "rows of the LDS image start on 16-byte boundaries");
        const float4* lds4 = reinterpret_cast<const float4*>(lds);
        const bool whole = (long)q0 + kBQ <= qend;
        if (whole) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wave * 8 + i, co = cb * BN + g * 32 + row;
                if (co >= a.Cout) continue;
                float4* orow4 = reinterpret_cast<float4*>(outn + (long)co * a.out_plane + q0);
#pragma unroll
                for (int h = 0; h < 2; ++h) orow4[h * 64 + lane] = lds4[row * (kFlatRow / 4) + h * 64 + lane];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = wave * 8 + i, co = cb * BN + g * 32 + row;
                if (co >= a.Cout) continue;
                float* orow = outn + (long)co * a.out_plane;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p4 = (h * 64 + lane) * 4;
                    const float4 v = lds4[row * (kFlatRow / 4) + h * 64 + lane];
                    if (q0 + p4 + 3 < qend) *reinterpret_cast<float4*>(orow + q0 + p4) = v;
                    else {                                         // ragged end of a plane whose size is not a multiple of 4
                        if (q0 + p4 < qend) orow[q0 + p4] = v.x;
                        if (q0 + p4 + 1 < qend) orow[q0 + p4 + 1] = v.y;
                        if (q0 + p4 + 2 < qend) orow[q0 + p4 + 2] = v.z;
                    }
                }
            }
        }
        if (tq && g == 0) tq[5] = __builtin_amdgcn_s_memtime();
    }
}

}  // namespace ttcconv
