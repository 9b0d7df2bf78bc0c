// ttc_calibrate_precision: choose, per conv layer of the 16-bit engine, how many of the three split products it multiplies -- for the CALLER's
// weights and the CALLER's windows.
//
// The 16-bit engine stores every conv operand as hi + lo 16-bit halves and multiplies lo*w_hi + hi*w_lo + hi*w_hi per layer (3.3 x the
// algorithmic matrix work, include/ttc.h `one_term_layers` / `two_term_layers`).  Whether a layer NEEDS all three is a property of the
// weights: GroupNorm (src/train/src/model.py:100-121) divides by the per-group standard deviation of the conv output, so a layer whose
// outputs are nearly constant in a group amplifies a 2^-11 operand error by 1 / std.  Every round so far decided "all layers, three
// products" on seeded stand-in weights (the trained checkpoint is absent from the reference checkout); a user WITH the trained weights
// could not find out what their model needs.  This entry measures it: the sample windows go through an fp32 reference context and through
// candidate maps on the 16-bit context, the cheapest map whose max |dprob| against the fp32 engine stays inside `budget` is applied.
#include <algorithm>
#include <cmath>

#include "ttc_internal.h"

ttc_status model_set_terms(ttc_ctx* c, uint32_t one, uint32_t two);    // model.hip

namespace {

// max |a - b| over n floats -> *out (as the bit pattern of a non-negative float: integer max is float max there); NaN counts as +inf
__global__ void k_max_abs_diff(const float* __restrict__ a, const float* __restrict__ b, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d != d) ? INFINITY : fmaxf(m, d);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

}  // namespace

extern "C" ttc_status ttc_calibrate_precision(ttc_ctx* c, ttc_ctx* ref, const float* d_windows, int32_t n, float budget,
                                              ttc_precision_report* rep, void* stream) {
    if (!c) return TTC_ERR_ARG;
    if (!ref || !d_windows || !rep) return c->fail(TTC_ERR_ARG, "calibrate_precision: null argument");
    if (!c->half()) return c->fail(TTC_ERR_ARG, "calibrate_precision: the context to calibrate must run the 16-bit engine (precision 2 = fp16 / 3 = bf16)");
    if (ref->cfg.precision != 0) return c->fail(TTC_ERR_ARG, "calibrate_precision: the reference context must be an fp32 context (precision 0)");
    if (ref->device != c->device) return c->fail(TTC_ERR_ARG, "calibrate_precision: both contexts must live on one device");
    if (ref->cfg.win_in != c->cfg.win_in || ref->cfg.win_rows != c->cfg.win_rows || ref->cfg.length != c->cfg.length)
        return c->fail(TTC_ERR_ARG, "calibrate_precision: the two contexts differ in window geometry / length");
    if (!c->have_model || !ref->have_model) return c->fail(TTC_ERR_STATE, "calibrate_precision: ttc_load_weights has not been called on both contexts");
    if (n < 1 || n > c->cfg.max_windows || n > ref->cfg.max_windows) return c->fail(TTC_ERR_ARG, "calibrate_precision: window count exceeds max_windows");
    if (!(budget >= 0.f)) return c->fail(TTC_ERR_ARG, "calibrate_precision: budget must be >= 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int W = c->cfg.win_in, H = c->cfg.win_rows ? c->cfg.win_rows : W, L = c->cfg.length;
    const long per = (long)n * (H - 14) * (W - 14);
    float* p_ref = static_cast<float*>(c->scratch_buf("cal_ref", sizeof(float) * per));
    float* p_out = static_cast<float*>(c->scratch_buf("cal_out", sizeof(float) * per));
    unsigned* d_max = static_cast<unsigned*>(c->scratch_buf("cal_max", 256));
    if (!p_ref || !p_out || !d_max) return c->fail(TTC_ERR_NOMEM, "calibrate_precision scratch");
    {   // the reference: the fp32 engine on the same windows (its own error against the fp64 graph is <= 5e-5, tests/test_gpu_model.py)
        ttc_status st = ttc_forward_windows(ref, d_windows, n, p_ref, stream);
        if (st != TTC_OK) return c->fail(st, std::string("calibrate_precision: reference forward: ") + ref->err);
    }
    const uint32_t saved_one = c->cfg.one_term_layers, saved_two = c->cfg.two_term_layers;
    const uint32_t ds_bits = saved_one & ~0x3FFu;                 // the DSen2 bits (10..15) are not calibrated here: kept as configured
    int trials = 0;
    auto run = [&](uint32_t one, uint32_t two, float* err) -> ttc_status {
        TTC_CHECK(model_set_terms(c, one | ds_bits, two));
        TTC_CHECK(ttc_forward_windows(c, d_windows, n, p_out, stream));
        TTC_HIP(c, hipMemsetAsync(d_max, 0, sizeof(unsigned), s));
        hipLaunchKernelGGL(k_max_abs_diff, dim3(256), dim3(256), 0, s, p_ref, p_out, per, d_max);
        unsigned bits = 0;
        TTC_HIP(c, hipMemcpyAsync(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        TTC_HIP(c, hipStreamSynchronize(s));
        memcpy(err, &bits, 4);
        ++trials;
        return TTC_OK;
    };
    auto fail_restore = [&](ttc_status st) { (void)model_set_terms(c, saved_one, saved_two); return st; };
    // matrix work per layer (multiply-accumulates per window, 9 taps dropped: a common factor): what a dropped product saves
    const int c1 = W / 2 - 2, c2 = c1 / 2 - 2, u2 = 2 * c2, u3 = 2 * u2, o = u3 - 2;
    const double hw = (double)H / W;                               // rectangular windows scale every plane alike
    const double cost[TTC_CAL_LAYERS] = {49.0 * 64 * W * W * 2 * L * hw, 49.0 * 32 * W * W * 2 * L * hw, 17.0 * 64 * W * W * hw, 128.0 * 64 * W * W * hw,
                                         64.0 * 128 * c1 * c1 * hw, 128.0 * 256 * c2 * c2 * hw, 256.0 * 128 * u2 * u2 * hw, 256.0 * 128 * u2 * u2 * hw,
                                         128.0 * 64 * u3 * u3 * hw, 128.0 * 64 * o * o * hw};
    const bool can_two = c->cfg.precision == 2;                    // the two-product kernels exist for the fp16 engine
    memset(rep, 0, sizeof(*rep));
    rep->budget = budget;
    ttc_status st;
    if ((st = run(0, 0, &rep->dprob_all_three)) != TTC_OK) return fail_restore(st);
    // every layer ON ITS OWN with one / two products, the others on three: the table a user reads to see which layers their weights protect
    for (int l = 0; l < TTC_CAL_LAYERS; ++l) {
        if ((st = run(1u << l, 0, &rep->layer_dprob_one[l])) != TTC_OK) return fail_restore(st);
        rep->layer_dprob_two[l] = NAN;
        if (can_two && (st = run(0, 1u << l, &rep->layer_dprob_two[l])) != TTC_OK) return fail_restore(st);
    }
    // greedy, most matrix work first: a layer takes the cheapest form that keeps the WHOLE map inside the budget (errors of different layers do
    // not add linearly -- every candidate is a full forward of the map as it would run)
    int order[TTC_CAL_LAYERS];
    for (int l = 0; l < TTC_CAL_LAYERS; ++l) order[l] = l;
    std::sort(order, order + TTC_CAL_LAYERS, [&](int a, int b) { return cost[a] > cost[b]; });
    uint32_t one = 0, two = 0;
    float cur = rep->dprob_all_three;
    for (int k = 0; k < TTC_CAL_LAYERS; ++k) {
        const int l = order[k];
        float e = INFINITY;
        if (rep->layer_dprob_one[l] <= budget) {                   // alone it already breaks the budget -> not worth a forward
            if ((st = run(one | (1u << l), two, &e)) != TTC_OK) return fail_restore(st);
            if (e <= budget) { one |= 1u << l; cur = e; continue; }
        }
        if (can_two && rep->layer_dprob_two[l] <= budget) {
            if ((st = run(one, two | (1u << l), &e)) != TTC_OK) return fail_restore(st);
            if (e <= budget) { two |= 1u << l; cur = e; }
        }
    }
    if ((st = run(one, two, &cur)) != TTC_OK) return fail_restore(st);      // leaves the chosen map applied; the number reported is of THIS state
    double work = 0, full = 0;
    for (int l = 0; l < TTC_CAL_LAYERS; ++l) {
        const int t = ((one >> l) & 1u) ? 1 : (((two >> l) & 1u) ? 2 : 3);
        work += cost[l] * t; full += cost[l] * 3;
    }
    rep->one_term_layers = one;
    rep->two_term_layers = two;
    rep->max_dprob = cur;
    rep->within_budget = cur <= budget ? 1 : 0;                     // 0: even three products everywhere exceed the budget on these windows
    rep->matrix_work_ratio = work / full;
    rep->trials = trials;
    rep->n_windows = n;
    return TTC_OK;
}
