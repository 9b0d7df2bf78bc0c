// Device helpers shared by the 16-bit conv engine (conv3x3_h16.hip) and the kernels that produce / consume its
// channel-blocked activations: element traits (fp16 / bf16), hi + lo splitting, 16-byte K-vector loads and stores.
#pragma once
#include "conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// BF: 0 = fp16 pair, 1 = bf16 pair
template <int BF> struct Elem;
template <> struct Elem<0> {
    using v8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b, unsigned& lo2) {
        // finite overflow saturates (the fp16 pair then stays finite); NaN must survive like it does on the fp32 engine --
        // fmaxf(NaN, x) returns x, so the clamp alone would turn it into -65504
        a = a != a ? a : fminf(fmaxf(a, -65504.f), 65504.f); b = b != b ? b : fminf(fmaxf(b, -65504.f), 65504.f);
        f16x2 h; h[0] = (_Float16)a; h[1] = (_Float16)b;
        f16x2 l; l[0] = (_Float16)(a - (float)h[0]); l[1] = (_Float16)(b - (float)h[1]);
        lo2 = __builtin_bit_cast(unsigned, l);
        return __builtin_bit_cast(unsigned, h);
    }
    static __device__ __forceinline__ void unpack2(unsigned u, float& a, float& b) {
        const f16x2 h = __builtin_bit_cast(f16x2, u); a = (float)h[0]; b = (float)h[1];
    }
};
template <> struct Elem<1> {
    using v8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b, unsigned& lo2) {
        bf16x2 h; h[0] = (__bf16)a; h[1] = (__bf16)b;
        bf16x2 l; l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
        lo2 = __builtin_bit_cast(unsigned, l);
        return __builtin_bit_cast(unsigned, h);
    }
    static __device__ __forceinline__ void unpack2(unsigned u, float& a, float& b) {
        const bf16x2 h = __builtin_bit_cast(bf16x2, u); a = (float)h[0]; b = (float)h[1];
    }
};


// Raw (pre-GroupNorm) conv outputs of the 16-bit engine are EXACT fp32 values stored channel-blocked as a plane of top and a
// plane of bottom 16-bit halves ([n][C8][P][8] each, one 16-byte vector per (channel block, position) and plane): the conv
// epilogue stores them straight from its accumulators, the elementwise consumers read 16-byte vectors.
__device__ __forceinline__ unsigned raw_top2(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ unsigned raw_bot2(float a, float b) { return (__float_as_uint(a) & 0xffffu) | (__float_as_uint(b) << 16); }
struct Raw16 { const uint4* top; const uint4* bot; };     // [n][C8][P] units each; bot = top + n * C8 * P for a launch over n sequences
__device__ __forceinline__ void raw_load8(const uint4* top, const uint4* bot, long u, float (&v)[8]) {
    const uint4 h = top[u], l = bot[u];
    v[0] = __uint_as_float((h.x << 16) | (l.x & 0xffffu)); v[1] = __uint_as_float((h.x & 0xffff0000u) | (l.x >> 16));
    v[2] = __uint_as_float((h.y << 16) | (l.y & 0xffffu)); v[3] = __uint_as_float((h.y & 0xffff0000u) | (l.y >> 16));
    v[4] = __uint_as_float((h.z << 16) | (l.z & 0xffffu)); v[5] = __uint_as_float((h.z & 0xffff0000u) | (l.z >> 16));
    v[6] = __uint_as_float((h.w << 16) | (l.w & 0xffffu)); v[7] = __uint_as_float((h.w & 0xffff0000u) | (l.w >> 16));
}

// one channel block (8 channels) of one position: x = hi + lo
template <int BF>
__device__ __forceinline__ void b16_store8(uint4* hi, uint4* lo, long u, const float (&v)[8]) {
    using E = Elem<BF>;
    uint4 h, l;
    h.x = E::pack2(v[0], v[1], l.x); h.y = E::pack2(v[2], v[3], l.y);
    h.z = E::pack2(v[4], v[5], l.z); h.w = E::pack2(v[6], v[7], l.w);
    hi[u] = h;
    if (lo) lo[u] = l;
}
template <int BF>
__device__ __forceinline__ void b16_load8(const uint4* hi, const uint4* lo, long u, float (&v)[8]) {
    using E = Elem<BF>;
    const uint4 h = hi[u];
    E::unpack2(h.x, v[0], v[1]); E::unpack2(h.y, v[2], v[3]); E::unpack2(h.z, v[4], v[5]); E::unpack2(h.w, v[6], v[7]);
    if (lo) {
        const uint4 l = lo[u];
        float t[8];
        E::unpack2(l.x, t[0], t[1]); E::unpack2(l.y, t[2], t[3]); E::unpack2(l.z, t[4], t[5]); E::unpack2(l.w, t[6], t[7]);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
}

// fp32 planar images [img][C][PP] -> channel-blocked hi / lo [img][C8][PP][8] (pad channels = 0)
template <int BF>
__global__ void k_planar_to_b16(const float* __restrict__ src, int C, long PP, int C8, uint4* __restrict__ hi, uint4* __restrict__ lo) {
    const long img = blockIdx.y;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= PP) return;
    const float* sp = src + img * C * PP + p;
    for (int k = 0; k < C8; ++k) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (8 * k + j < C) ? sp[(long)(8 * k + j) * PP] : 0.0f;
        b16_store8<BF>(hi, lo, (img * C8 + k) * PP + p, v);
    }
}

