// 8-bit-per-pass radix select on float keys, batched over problems, with DEVICE-side ranks
// (shared by gapfill.hip and codecs.hip).  A "source" functor supplies value(q, p) and the element count.
#pragma once
#include "ttc_internal.h"

namespace ttcsel {

__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
struct SelState { unsigned prefix, mask; long long k; };

template <class SRC>
__global__ void k_sel_hist(SRC s, const SelState* __restrict__ st, int shift, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    const int q = blockIdx.y;
    h[threadIdx.x] = 0;
    __syncthreads();
    const SelState ss = st[q];
    const int n = s.count();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        float v;
        if (!s.get(q, p, v)) continue;
        const unsigned k = fkey(v);
        if ((k & ss.mask) == ss.prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[q * 256 + threadIdx.x], h[threadIdx.x]);
}
static __global__ void k_sel_pick(SelState* __restrict__ st, int shift, unsigned* __restrict__ hist) {
    // one wave per problem: lane l owns bins 4l..4l+3; wave prefix sum finds the bin holding rank k
    const int q = blockIdx.x, lane = threadIdx.x;
    unsigned c[4];
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = hist[q * 256 + 4 * lane + j]; mine += c[j]; hist[q * 256 + 4 * lane + j] = 0; }
    unsigned incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const long long excl = (long long)incl - mine;
    SelState ss = st[q];
    const long long k = ss.k;
    const bool here = k >= excl && k < (long long)incl;
    const unsigned long long m = __ballot(here);
    const int owner = m ? __ffsll((long long)m) - 1 : 63;       // ranks past the last element (empty set) fall in the last bin
    if (lane == owner) {
        long long r = k - excl;
        int b = 0;
        for (; b < 3; ++b) { if (r < (long long)c[b]) break; r -= c[b]; }
        ss.prefix |= (unsigned)(4 * lane + b) << shift; ss.mask |= 255u << shift; ss.k = m ? r : 0;
        st[q] = ss;
    }
}
// ranks from a DEVICE count: mode 0 -> the two middle order statistics (median); mode 1 -> floor / floor+1 of
// numpy's linear-interpolation position pct/100 * (n - 1).  n = *n_ptr, or n_total - *n_ptr when complement.
struct PctList { double pct[8]; };
static __global__ void k_sel_init(SelState* __restrict__ st, int nprob, const int* __restrict__ n_ptr, int n_total, int complement,
                           int mode, PctList pl) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nprob) return;
    long long n = complement ? (long long)n_total - *n_ptr : *n_ptr;
    SelState ss; ss.prefix = 0; ss.mask = 0; ss.k = 0;
    if (n > 0) {
        if (mode == 0) ss.k = (q & 1) ? n / 2 : (n - 1) / 2;
        else {
            const double pos = pl.pct[q >> 1] / 100.0 * (double)(n - 1);
            long long lo = (long long)floor(pos) + (q & 1);
            ss.k = lo > n - 1 ? n - 1 : lo;
        }
    }
    st[q] = ss;
}
template <class SRC>
inline hipError_t radix_select(SRC src, SelState* st, unsigned* hist, int nprob, hipStream_t s) {
    for (int shift = 24; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL((k_sel_hist<SRC>), dim3(96, nprob), dim3(256), 0, s, src, st, shift, hist);
        hipLaunchKernelGGL(k_sel_pick, dim3(nprob), dim3(64), 0, s, st, shift, hist);
    }
    return hipGetLastError();
}

}  // namespace ttcsel
