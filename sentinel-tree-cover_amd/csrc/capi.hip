// extern "C" entry points of libttc_hip.so (see include/ttc.h for the contract).
#include <algorithm>
#include <cstddef>

#include "ttc_internal.h"

// tile.hip / mosaic.hip / dsen2.hip
ttc_status tile_process_subtiles(ttc_ctx* c, const float* d_s2, int T, int X, int Y, const float* h_wmat,
                                 const int32_t* h_keep, const float* d_interp, const float* d_s1, const float* d_dem,
                                 const double* h_min, const double* h_max, int size, int n_dates_ok, float* d_windows,
                                 float* d_windows_raw, hipStream_t s);
ttc_status tile_process_subtiles_dev(ttc_ctx* c, float* d_s2, int T, int X, int Y, const int32_t* d_dates, const float* d_interp,
                                     const float* d_s1, const float* d_dem, const double* h_min, const double* h_max, int size,
                                     float* d_windows, float* d_windows_raw, bool stop_after_inputs, hipStream_t s);
ttc_status tile_missing_counts(ttc_ctx* c, const float* d_s2, int T, int X, int Y, int32_t* d_counts, hipStream_t s);
ttc_status tile_fix_missing(ttc_ctx* c, float* d_s2, int T, int X, int Y, int do_nan, int do_zero_one, hipStream_t s);
ttc_status mosaic_run(ttc_ctx* c, const float* d_windows, int n, const int32_t* h_xy, int size, int rows, int cols,
                      uint8_t* d_u8, float* d_f32, hipStream_t s);
ttc_status dsen2_forward(ttc_ctx* c, const float* d_in, const float* d_bil, int n, int H, int W, float* d_out,
                         hipStream_t s);
ttc_status dsen2_tile(ttc_ctx* c, float* d_s2, int T, int X, int Y, int quirks, int ws, int cs, hipStream_t s);
ttc_status tile_smooth_strip(ttc_ctx* c, const float* d_s2, int T, int X, int Y, const float* h_wmat, float* d_out, hipStream_t s);
ttc_status upsample_20m(ttc_ctx* c, const float* d10, const float* d20, int T, int h, int w, float* d_out, hipStream_t s);
ttc_status decode_upsample_u16(ttc_ctx* c, const uint16_t* d10, const uint16_t* d20, int T, int h, int w, const AdjustMap* am10, float* d_out,
                               hipStream_t s);

ttc_status gapfill_feather(ttc_ctx* c, const float* d_mask, int T, int X, int Y, int closing, int clip, float* d_w, hipStream_t s);
ttc_status gapfill_aligned_mosaic(ttc_ctx* c, const float* d_tiles, float* d_w, int T, int X, int Y, float* d_mosaic, hipStream_t s);
ttc_status gapfill_remove_clouds(ttc_ctx* c, float* d_tiles, const float* d_probs, const uint8_t* d_pfcps, int T, int X, int Y,
                                 ttc_sampler_fn sampler, void* user, float* d_interp, float* d_mosaic_out, int32_t* h_to_remove,
                                 int32_t* n_to_remove, hipStream_t s);

ttc_status codec_u16_to_f32(ttc_ctx* c, const uint16_t* d_in, int64_t n, float* d_out, hipStream_t s);
ttc_status codec_f32_to_u16(ttc_ctx* c, const float* d_in, int64_t n, uint16_t* d_out, hipStream_t s);
ttc_status codec_s1_to_db(ttc_ctx* c, const uint16_t* d_u16, int T, int X, int Y, float* d_out, hipStream_t s, const AdjustMap* am = nullptr);
ttc_status codec_adjust_shape(ttc_ctx* c, const float* d_in, int T, int n1, int n2, int C, int width, int height, float* d_out, hipStream_t s);
ttc_status codec_f32_to_i16(ttc_ctx* c, const float* d_in, int64_t n, float precision, int16_t* d_out, hipStream_t s);

ttc_status mosaic_features(ttc_ctx* c, const int16_t* d_feats, int n, const int32_t* h_xy, int size, int depth, int rows, int cols,
                           int16_t* d_out, hipStream_t s);

ttc_status clouds_identify(ttc_ctx* c, const float* img, int T, int X, int Y, const float* dem, const uint8_t* forest,
                           const uint8_t* urban_core, const uint8_t* urban_near, float* d_clouds, uint8_t* d_fcps, int debug_stage,
                           hipStream_t s);

ttc_status prep_sen2cor_clean(ttc_ctx* c, const float* d_clm20, int T, int w20, int h20, float* d_out, hipStream_t s);
ttc_status prep_median5(ttc_ctx* c, const float* d_in, int X, int Y, float* d_out, hipStream_t s);
ttc_status prep_snow(ttc_ctx* c, const float* d_s2, int T, int X, int Y, uint8_t* d_snow, int32_t* h_per_image, hipStream_t s);
ttc_status prep_merge_clm(ttc_ctx* c, float* d_cloudshad, float* d_clm, const uint8_t* d_fcps, int64_t n, hipStream_t s);
ttc_status prep_count_positive(ttc_ctx* c, const float* d_a, int T, int npix, float eq, int32_t* h_counts, hipStream_t s);
ttc_status prep_clip01(ttc_ctx* c, float* d_a, int64_t n, hipStream_t s);
ttc_status prep_divide(ttc_ctx* c, float* d_a, int64_t n, float divisor, hipStream_t s);

void h16_set_knob(int which, int value);      // conv3x3_h16.hip

static void flush_timing(ttc_ctx* c) {
    for (auto& p : c->timing.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
            auto& r = c->timing.recs[p.first];
            r.ms += ms; r.n += 1;
        }
        (void)hipEventDestroy(p.second.first);
        (void)hipEventDestroy(p.second.second);
    }
    c->timing.pending.clear();
}

// bytes of a guard zone that no longer hold the pattern; first[0] = smallest offending offset
__global__ void k_guard_scan(const unsigned char* __restrict__ z, size_t n, unsigned long long* __restrict__ bad, unsigned long long* __restrict__ first) {
    unsigned long long mine = 0, lo = ~0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (z[i] != 0xA5) { ++mine; lo = lo < i ? lo : i; }
    if (mine) { atomicAdd(bad, mine); atomicMin(first, lo); }
}

extern "C" {

ttc_status ttc_debug_check_guards(ttc_ctx* c, int64_t* n_bad, char* msg, size_t cap) {
    if (!c || !n_bad) return TTC_ERR_ARG;
    *n_bad = 0;
    if (msg && cap) msg[0] = 0;
    const size_t G = ttc_ctx::guard_bytes();
    if (!G) return c->fail(TTC_ERR_STATE, "check_guards: the library runs without guard zones (set TTC_GUARD=<KiB> before the first context)");
    TTC_HIP(c, hipSetDevice(c->device));
    TTC_HIP(c, hipDeviceSynchronize());
    unsigned long long* d = nullptr;
    TTC_HIP(c, hipMalloc(&d, 16));
    std::string first_msg;
    for (auto& kv : c->guarded) {
        const auto& g = kv.second;
        const size_t user = (g.user_bytes + 255) & ~(size_t)255;
        struct Z { const char* what; const char* p; size_t n; } zones[2] = {{"before", g.base, G}, {"after", g.base + G + g.user_bytes, user - g.user_bytes + G}};
        for (auto& z : zones) {
            unsigned long long h[2] = {0ull, ~0ull};
            TTC_HIP(c, hipMemcpy(d, h, 16, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_guard_scan, dim3(64), dim3(256), 0, 0, reinterpret_cast<const unsigned char*>(z.p), z.n, d, d + 1);
            TTC_HIP(c, hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            if (h[0]) {
                *n_bad += (int64_t)h[0];
                if (first_msg.empty()) {
                    char b[320];
                    snprintf(b, sizeof b, "%s (%zu bytes): %llu byte(s) overwritten %s the buffer, first at %s%llu", g.name.c_str(), g.user_bytes, h[0], z.what,
                             z.what[0] == 'a' ? "end+" : "start-", z.what[0] == 'a' ? h[1] : (unsigned long long)G - h[1]);
                    first_msg = b;
                }
            }
        }
    }
    (void)hipFree(d);
    if (msg && cap) snprintf(msg, cap, "%s", first_msg.c_str());
    return TTC_OK;
}

const char* ttc_version(void) { return "ttc-hip 0.1 (gfx950)"; }

size_t ttc_config_size(void) { return sizeof(ttc_config); }

// ttc_config grows at its END from release to release: a caller built against an older header hands over a shorter struct.  _v2 takes the
// caller's sizeof(ttc_config): missing trailing fields read as zero (every field added so far defaults to 0), a LONGER struct than this
// library knows is refused (it may carry a request the library would silently ignore).
ttc_status ttc_create_v2(ttc_ctx** out, int32_t device, const ttc_config* cfg, size_t cfg_size) {
    if (!out || !cfg) return TTC_ERR_ARG;
    *out = nullptr;
    // the first nine fields (up to win_rows) are the round-1 struct: nothing shorter was ever shipped
    if (cfg_size < offsetof(ttc_config, one_term_layers) || cfg_size > sizeof(ttc_config)) return TTC_ERR_ARG;
    ttc_config full;
    memset(&full, 0, sizeof(full));
    memcpy(&full, cfg, cfg_size);
    return ttc_create(out, device, &full);
}

ttc_status ttc_create(ttc_ctx** out, int32_t device, const ttc_config* cfg) {
    if (!out || !cfg) return TTC_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return TTC_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return TTC_ERR_HIP;
    ttc_ctx* c = new ttc_ctx();
    c->cfg = *cfg;
    c->device = device;
    *out = c;                       // returned even on failure so the caller can read ttc_last_error
    if (cfg->precision != 0 && cfg->precision != 2 && cfg->precision != 3)
        return c->fail(TTC_ERR_ARG, "precision: 0 (exact fp32 MFMA), 2 (fp16) / 3 (bf16) hi + lo pairs on the 16-bit engine; 1 and 4 named the retired "
                                    "bf16x3 / fp32-blocked engines (csrc/experiments/)");
    if (cfg->max_windows < 1 || cfg->length < 1) return c->fail(TTC_ERR_ARG, "max_windows and length must be >= 1");
    if (cfg->fp32_conv_form < 0 || cfg->fp32_conv_form > 2)
        return c->fail(TTC_ERR_ARG, "fp32_conv_form: 0 (Winograd F(4x4,3x3) where it applies), 1 (F(2x2,3x3) at most), 2 (direct only)");
    if (cfg->dsen2_precision != 0 && cfg->dsen2_precision != 2 && cfg->dsen2_precision != 3)
        return c->fail(TTC_ERR_ARG, "dsen2_precision: 0 (the context's precision), 2 (fp16) / 3 (bf16) hi + lo pairs on the 16-bit engine");
    return model_alloc(c);
}

void ttc_destroy(ttc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    flush_timing(c);
    for (void* p : c->allocs) c->guarded_free(p);
    for (auto& kv : c->scratch) c->guarded_free(kv.second.first);
    for (auto& kv : c->pinned) (void)hipHostFree(kv.second.first);
    for (hipEvent_t e : c->wmat_events) if (e) (void)hipEventDestroy(e);
    delete c;
}

const char* ttc_last_error(const ttc_ctx* c) { return c ? c->err.c_str() : "null context"; }
size_t ttc_device_bytes(const ttc_ctx* c) { return c ? c->dev_bytes : 0; }

ttc_status ttc_load_weights(ttc_ctx* c, const ttc_tensor* t, int32_t n) {
    if (!c || !t) return TTC_ERR_ARG;
    return model_load(c, t, n);
}

ttc_status ttc_load_dsen2_weights(ttc_ctx* c, const ttc_tensor* t, int32_t n) {
    if (!c || !t) return TTC_ERR_ARG;
    return dsen2_load(c, t, n);
}

ttc_status ttc_forward_windows(ttc_ctx* c, const float* d_in, int32_t n, float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    if (!d_in || !d_out) return c->fail(TTC_ERR_ARG, "null buffer");
    if (n <= 0 || n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "window count exceeds max_windows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TTC_CHECK(model_frames_from_nhwc(c, d_in, n, s));
    return model_forward_frames(c, n, d_out, s, FRAMES_PLANAR);
}

ttc_status ttc_forward_taps(ttc_ctx* c, const float* d_in, int32_t n, float* d_out, float* d_early, float* d_late, void* stream) {
    if (!c) return TTC_ERR_ARG;
    if (!d_in || !d_out) return c->fail(TTC_ERR_ARG, "null buffer");
    if (n <= 0 || n > c->cfg.max_windows) return c->fail(TTC_ERR_ARG, "window count exceeds max_windows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TTC_CHECK(model_frames_from_nhwc(c, d_in, n, s));
    // Feature export (job.py:1429-1445) hands the LATE features out as int16 thousandths; the F(4x4) form's rounding on the deep U-Net blocks
    // (raw outputs <= 2e-4 of the fp64 graph on values of magnitude 13) is a visible share of that quantum, so a forward whose late tap is
    // requested runs the eight blocks in the F(2x2) form (<= 5e-5: round 4's tolerances) -- ~1.3 ms more per 36-window forward, on the export
    // path only.  The probabilities returned by this call come from the same forward.
    int saved[8];
    const bool calm = d_late && c->cfg.precision == 0;
    for (int b = 0; b < 8; ++b) { saved[b] = c->w_block[b].form; if (calm && saved[b] == 0) c->w_block[b].form = 1; }
    ttc_status st = model_forward_frames(c, n, d_out, s, FRAMES_PLANAR);
    for (int b = 0; b < 8; ++b) c->w_block[b].form = saved[b];
    TTC_CHECK(st);
    return model_taps(c, n, d_early, d_late, s);
}

ttc_status ttc_mosaic_features(ttc_ctx* c, const int16_t* d_feats, int32_t n, const int32_t* h_xy, int32_t size, int32_t depth,
                               int32_t out_rows, int32_t out_cols, int16_t* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return mosaic_features(c, d_feats, n, h_xy, size, depth, out_rows, out_cols, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_border_subtiles(ttc_ctx* c, const float* d_s2, const float* d_s1, const float* d_dem, int32_t X,
                               const int32_t* h_rows, int32_t n, const float* h_min, const float* h_max, int32_t hist_align,
                               int32_t n_dates_ok, float* d_preds, float* h_stats, int32_t* h_applied, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return reseg_border_subtiles(c, d_s2, d_s1, d_dem, X, h_rows, n, h_min, h_max, hist_align, n_dates_ok, d_preds, h_stats,
                                 h_applied, static_cast<hipStream_t>(stream));
}

ttc_status ttc_superresolve_windows(ttc_ctx* c, float* d_arr, int32_t T, int32_t X, int32_t Y, int32_t channels, int32_t wsize,
                                    int32_t quirks, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return dsen2_tile(c, d_arr, T, X, Y, quirks, wsize, channels, static_cast<hipStream_t>(stream));
}

ttc_status ttc_smooth_strip(ttc_ctx* c, const float* d_s2, int32_t T, int32_t X, int32_t Y, const float* h_wmat, float* d_out,
                            void* stream) {
    if (!c) return TTC_ERR_ARG;
    return tile_smooth_strip(c, d_s2, T, X, Y, h_wmat, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_reseg_mosaic(ttc_ctx* c, const float* d_preds, const ttc_reseg_window* h_wins, int32_t n, const float* d_weights,
                            const double* d_ramps, int32_t X, int32_t Y, float* d_out, float* d_sums, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return reseg_mosaic(c, d_preds, h_wins, n, d_weights, d_ramps, X, Y, d_out, d_sums, static_cast<hipStream_t>(stream));
}

ttc_status ttc_seam_adjust(ttc_ctx* c, float* d_preds, int32_t n, int32_t rows, int32_t cols, float* h_stats, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return reseg_seam_adjust(c, d_preds, n, rows, cols, h_stats, static_cast<hipStream_t>(stream));
}

ttc_status ttc_float_to_int16(ttc_ctx* c, const float* d_in, int64_t n, float precision, int16_t* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return codec_f32_to_i16(c, d_in, n, precision, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_process_subtiles(ttc_ctx* c, const float* d_s2, int32_t T, int32_t X, int32_t Y, const float* h_wmat,
                                const int32_t* h_keep, const float* d_interp, const float* d_s1, const float* d_dem,
                                const double* h_min, const double* h_max, int32_t size, int32_t n_dates_ok,
                                float* d_windows, float* d_windows_raw, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return tile_process_subtiles(c, d_s2, T, X, Y, h_wmat, h_keep, d_interp, d_s1, d_dem, h_min, h_max, size, n_dates_ok,
                                 d_windows, d_windows_raw, static_cast<hipStream_t>(stream));
}

ttc_status ttc_tile_missing_counts(ttc_ctx* c, const float* d_s2, int32_t T, int32_t X, int32_t Y, int32_t* d_counts,
                                   void* stream) {
    if (!c) return TTC_ERR_ARG;
    return tile_missing_counts(c, d_s2, T, X, Y, d_counts, static_cast<hipStream_t>(stream));
}

ttc_status ttc_tile_fix_missing(ttc_ctx* c, float* d_s2, int32_t T, int32_t X, int32_t Y, int32_t do_nan,
                                int32_t do_zero_one, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return tile_fix_missing(c, d_s2, T, X, Y, do_nan, do_zero_one, static_cast<hipStream_t>(stream));
}

ttc_status ttc_feather(ttc_ctx* c, const float* d_mask, int32_t T, int32_t X, int32_t Y, int32_t closing, int32_t clip,
                       float* d_w, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return gapfill_feather(c, d_mask, T, X, Y, closing, clip, d_w, static_cast<hipStream_t>(stream));
}

ttc_status ttc_aligned_mosaic(ttc_ctx* c, const float* d_tiles, float* d_w, int32_t T, int32_t X, int32_t Y, float* d_mosaic,
                              void* stream) {
    if (!c) return TTC_ERR_ARG;
    return gapfill_aligned_mosaic(c, d_tiles, d_w, T, X, Y, d_mosaic, static_cast<hipStream_t>(stream));
}

ttc_status ttc_remove_cloud_and_shadows(ttc_ctx* c, float* d_tiles, const float* d_probs, const uint8_t* d_pfcps, int32_t T,
                                        int32_t X, int32_t Y, ttc_sampler_fn sampler, void* user, float* d_interp,
                                        float* d_mosaic, int32_t* h_to_remove, int32_t* n_to_remove, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return gapfill_remove_clouds(c, d_tiles, d_probs, d_pfcps, T, X, Y, sampler, user, d_interp, d_mosaic, h_to_remove,
                                 n_to_remove, static_cast<hipStream_t>(stream));
}

// the body of ttc_predict_tile / ttc_predict_tile_shaped: X, Y = the tile's grid (2 x the 20 m stack's, job.py:716-717); am10 / am_s1 map it
// onto the 10 m and Sentinel-1 arrays as stored (null = same shape); d_dem / d_dem_m / d_mask already have the tile's shape
static ttc_status predict_tile_impl(ttc_ctx* c, const uint16_t* d_s2_10, const uint16_t* d_s2_20, const uint16_t* d_s1, const float* d_dem,
                                    const float* d_dem_m, const float* d_mask, const int32_t* d_dates, int32_t T, int32_t X, int32_t Y,
                                    const AdjustMap* am10, const AdjustMap* am_s1, const double* h_min, const double* h_max, int32_t size,
                                    int32_t flags, uint8_t* d_out_u8, float* d_out_f32, float* d_model_in, int32_t* d_status,
                                    hipStream_t s) {
    const bool detect = (flags & TTC_TILE_DETECT) != 0, inputs_only = (flags & TTC_TILE_INPUTS_ONLY) != 0;
    if (!d_s2_10 || !d_s2_20 || !d_s1 || !d_dem || !d_dates || !h_min || !h_max || !d_status) return c->fail(TTC_ERR_ARG, "predict_tile: null argument");
    if (!detect && !d_mask) return c->fail(TTC_ERR_ARG, "predict_tile: a cloud / shadow mask is needed unless TTC_TILE_DETECT is set");
    if (detect && !d_dem_m) return c->fail(TTC_ERR_ARG, "predict_tile: TTC_TILE_DETECT needs the elevation in metres");
    if (!inputs_only && !d_out_u8) return c->fail(TTC_ERR_ARG, "predict_tile: no output raster");
    if (T < 1 || T > 32 || X < 2 || Y < 2 || (X & 1) || (Y & 1)) return c->fail(TTC_ERR_ARG, "predict_tile: T in [1,32], even X and Y");
    const int h = X / 2, w = Y / 2;
    const size_t npix = (size_t)X * Y;
    float* s1db = static_cast<float*>(c->scratch_buf("pt_s1", sizeof(float) * 12 * npix * 2));
    float* s2 = static_cast<float*>(c->scratch_buf("pt_s2", sizeof(float) * T * npix * 10));
    float* interp = static_cast<float*>(c->scratch_buf("pt_interp", sizeof(float) * T * npix));
    if (!s1db || !s2 || !interp) return c->fail(TTC_ERR_NOMEM, "predict_tile scratch");
    // window grid (job.py:1295-1316): origins of the 6 x 6 output windows, iteration order x-major
    std::vector<int32_t> xy;
    {
        const int gx = (X - size + 4) / 5, gy = (Y - size + 4) / 5;                      // ceil((X - size) / 5)
        if (X <= size || Y <= size || gx < 1 || gy < 1) return c->fail(TTC_ERR_ARG, "predict_tile: tile smaller than the window grid");
        std::vector<int> xs, ys;
        for (int v = 0; v < X - size; v += gx) xs.push_back(v);
        xs.push_back(X - size);
        for (int v = 0; v < Y - size; v += gy) ys.push_back(v);
        ys.push_back(Y - size);
        for (int x : xs) for (int y : ys) { xy.push_back(x); xy.push_back(y); }
    }
    const int n_win = (int)xy.size() / 2;
    float* windows = static_cast<float*>(c->scratch_buf("pt_windows", sizeof(float) * (size_t)n_win * size * size));
    float* windows_raw = static_cast<float*>(c->scratch_buf("pt_windows_raw", sizeof(float) * (size_t)n_win * size * size));
    if (!windows || !windows_raw) return c->fail(TTC_ERR_NOMEM, "predict_tile scratch");
    c->named["pt_windows"] = {windows, (size_t)n_win * size * size};           // what the reference np.save()s per window
    c->named["pt_windows_raw"] = {windows_raw, (size_t)n_win * size * size};   // before np.around / the bright-surface product
    TTC_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t) * 4, s));
    TTC_CHECK(codec_s1_to_db(c, d_s1, 12, X, Y, s1db, s, am_s1));                       // job.py:699-708 (+ adjust_shape :718)
    TTC_CHECK(decode_upsample_u16(c, d_s2_10, d_s2_20, T, h, w, am10, s2, s));          // tof_downloading.py:64-72 + job.py:720, :734-782, one pass
    const float* mask = d_mask;
    const uint8_t* pf = nullptr;
    if (detect) {                                                                       // cloud_removal.py:1215-1677
        float* clouds = static_cast<float*>(c->scratch_buf("pt_clouds", sizeof(float) * T * npix));
        uint8_t* fcps = static_cast<uint8_t*>(c->scratch_buf("pt_fcps", T * npix));
        if (!clouds || !fcps) return c->fail(TTC_ERR_NOMEM, "predict_tile scratch");
        TTC_CHECK(clouds_identify(c, s2, T, X, Y, d_dem_m, nullptr, nullptr, nullptr, clouds, fcps, 0, s));
        mask = clouds; pf = fcps;
    }
    c->want_planar_frames = d_model_in != nullptr;
    c->spec_status = d_status;                     // the speculative stages report into it instead of waiting for the host
    // cloud_removal.py:888-973; with spec_status set the blend also applies process_tile's final np.clip(sentinel2, 0, 1) (job.py:993)
    ttc_status st = gapfill_remove_clouds(c, s2, mask, pf, T, X, Y, nullptr, nullptr, interp, nullptr, nullptr, nullptr, s);
    if (st == TTC_OK && !(flags & TTC_TILE_NO_SUPERRES)) st = dsen2_tile(c, s2, T, X, Y, 1, 110, 10, s);                       // job.py:95-147
    if (st == TTC_OK) st = tile_process_subtiles_dev(c, s2, T, X, Y, d_dates, interp, s1db, d_dem, h_min, h_max, size, windows, windows_raw,
                                                     inputs_only, s);                                                         // job.py:1125-1483
    c->spec_status = nullptr;
    c->want_planar_frames = false;
    TTC_CHECK(st);
    if (d_model_in) {
        const size_t nfl = (size_t)n_win * (c->cfg.length + 1) * c->cfg.n_bands * (c->cfg.win_in + 2) * (c->cfg.win_in + 2);
        TTC_HIP(c, hipMemcpyAsync(d_model_in, c->frames, sizeof(float) * nfl, hipMemcpyDeviceToDevice, s));
    }
    if (inputs_only) return TTC_OK;
    int rows = 0, cols = 0;
    for (int i = 0; i < n_win; ++i) { rows = std::max(rows, xy[2 * i + 1] + size); cols = std::max(cols, xy[2 * i] + size); }
    return mosaic_run(c, windows, n_win, xy.data(), size, rows, cols, d_out_u8, d_out_f32, s);                                 // job.py:1515-1641
}

ttc_status ttc_predict_tile(ttc_ctx* c, const uint16_t* d_s2_10, const uint16_t* d_s2_20, const uint16_t* d_s1, const float* d_dem,
                            const float* d_dem_m, const float* d_mask, const int32_t* d_dates, int32_t T, int32_t X, int32_t Y,
                            const double* h_min, const double* h_max, int32_t size, int32_t flags, uint8_t* d_out_u8,
                            float* d_out_f32, float* d_model_in, int32_t* d_status, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return predict_tile_impl(c, d_s2_10, d_s2_20, d_s1, d_dem, d_dem_m, d_mask, d_dates, T, X, Y, nullptr, nullptr, h_min, h_max, size, flags,
                             d_out_u8, d_out_f32, d_model_in, d_status, static_cast<hipStream_t>(stream));
}

ttc_status ttc_predict_tile_shaped(ttc_ctx* c, const uint16_t* d_s2_10, const uint16_t* d_s2_20, const uint16_t* d_s1, const float* d_dem,
                                   const float* d_dem_m, const float* d_mask, const int32_t* d_dates, int32_t T,
                                   const ttc_tile_shapes* shp, const double* h_min, const double* h_max, int32_t size, int32_t flags,
                                   uint8_t* d_out_u8, float* d_out_f32, float* d_model_in, int32_t* d_status, void* stream) {
    if (!c) return TTC_ERR_ARG;
    if (!shp) return c->fail(TTC_ERR_ARG, "predict_tile_shaped: null shapes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (shp->s2_20[0] < 1 || shp->s2_20[1] < 1) return c->fail(TTC_ERR_ARG, "predict_tile_shaped: empty 20 m stack");
    const int X = 2 * shp->s2_20[0], Y = 2 * shp->s2_20[1];                  // job.py:716-717: the 20 m stack decides
    AdjustMap am10, am_s1, am_dem;
    auto bad = [&](const char* what, const int32_t* got) {
        char m[256];
        snprintf(m, sizeof m, "predict_tile_shaped: %s is %d x %d for a %d x %d tile -- adjust_shape (job.py:260-310) reconciles differences of 1 "
                 "or of an even number of pixels only (the reference raises on the rest)", what, got[0], got[1], X, Y);
        return c->fail(TTC_ERR_ARG, m);
    };
    if (!adjust_map(shp->s2_10[0], shp->s2_10[1], X, Y, &am10)) return bad("s2_10", shp->s2_10);
    if (!adjust_map(shp->s1[0], shp->s1[1], X, Y, &am_s1)) return bad("s1", shp->s1);
    if (!adjust_map(shp->dem[0], shp->dem[1], X, Y, &am_dem)) return bad("dem", shp->dem);
    // the cloud / shadow mask is not one of process_tile's files: it is made ON the tile's grid (identify_clouds_shadows, :839), so a mask of
    // another shape is a caller error, not something adjust_shape covers
    if (d_mask && !(flags & TTC_TILE_DETECT) && (shp->mask[0] != X || shp->mask[1] != Y)) {
        char m[200];
        snprintf(m, sizeof m, "predict_tile_shaped: the mask is %d x %d, the tile (2 x the 20 m stack) is %d x %d", shp->mask[0], shp->mask[1], X, Y);
        return c->fail(TTC_ERR_ARG, m);
    }
    const float *dem = d_dem, *dem_m = d_dem_m;
    if (am_dem.n1 != X || am_dem.n2 != Y) {                                  // dem = adjust_shape(median_filter(dem, 5), ...), :713, :721
        if (!d_dem) return c->fail(TTC_ERR_ARG, "predict_tile: null argument");
        float* a = static_cast<float*>(c->scratch_buf("pt_dem_adj", sizeof(float) * 2 * (size_t)X * Y));
        if (!a) return c->fail(TTC_ERR_NOMEM, "predict_tile scratch");
        TTC_CHECK(codec_adjust_shape(c, d_dem, 1, am_dem.n1, am_dem.n2, 1, X, Y, a, s));
        dem = a;
        if (d_dem_m) { TTC_CHECK(codec_adjust_shape(c, d_dem_m, 1, am_dem.n1, am_dem.n2, 1, X, Y, a + (size_t)X * Y, s)); dem_m = a + (size_t)X * Y; }
    }
    return predict_tile_impl(c, d_s2_10, d_s2_20, d_s1, dem, dem_m, d_mask, d_dates, T, X, Y, &am10, &am_s1, h_min, h_max, size, flags, d_out_u8,
                             d_out_f32, d_model_in, d_status, s);
}

ttc_status ttc_adjust_shape(ttc_ctx* c, const float* d_in, int32_t T, int32_t n1, int32_t n2, int32_t channels, int32_t width, int32_t height,
                            float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return codec_adjust_shape(c, d_in, T, n1, n2, channels, width, height, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_mosaic(ttc_ctx* c, const float* d_windows, int32_t n, const int32_t* h_xy, int32_t size,
                      int32_t out_rows, int32_t out_cols, uint8_t* d_out_u8, float* d_out_f32, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return mosaic_run(c, d_windows, n, h_xy, size, out_rows, out_cols, d_out_u8, d_out_f32,
                      static_cast<hipStream_t>(stream));
}

ttc_status ttc_dsen2_forward(ttc_ctx* c, const float* d_in, const float* d_bilinear, int32_t n, int32_t H, int32_t W,
                             float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return dsen2_forward(c, d_in, d_bilinear, n, H, W, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_superresolve_tile(ttc_ctx* c, float* d_s2, int32_t T, int32_t X, int32_t Y, int32_t quirks,
                                 void* stream) {
    if (!c) return TTC_ERR_ARG;
    return dsen2_tile(c, d_s2, T, X, Y, quirks, 110, 10, static_cast<hipStream_t>(stream));
}

ttc_status ttc_upsample_20m(ttc_ctx* c, const float* d_s2_10, const float* d_s2_20, int32_t T, int32_t h, int32_t w,
                            float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return upsample_20m(c, d_s2_10, d_s2_20, T, h, w, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_u16_to_float(ttc_ctx* c, const uint16_t* d_in, int64_t n, float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return codec_u16_to_f32(c, d_in, n, d_out, static_cast<hipStream_t>(stream));
}
ttc_status ttc_float_to_u16(ttc_ctx* c, const float* d_in, int64_t n, uint16_t* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return codec_f32_to_u16(c, d_in, n, d_out, static_cast<hipStream_t>(stream));
}
ttc_status ttc_s1_to_db(ttc_ctx* c, const uint16_t* d_u16, int32_t T, int32_t X, int32_t Y, float* d_out, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return codec_s1_to_db(c, d_u16, T, X, Y, d_out, static_cast<hipStream_t>(stream));
}

ttc_status ttc_identify_clouds_shadows(ttc_ctx* c, const float* d_img, int32_t T, int32_t X, int32_t Y, const float* d_dem,
                                       const uint8_t* d_forest, const uint8_t* d_urban_core, const uint8_t* d_urban_near,
                                       float* d_clouds, uint8_t* d_fcps, void* stream) {
    if (!c) return TTC_ERR_ARG;
    return clouds_identify(c, d_img, T, X, Y, d_dem, d_forest, d_urban_core, d_urban_near, d_clouds, d_fcps, c->clouds_debug_stage,
                           static_cast<hipStream_t>(stream));
}

#define TTC_S(x) static_cast<hipStream_t>(x)
ttc_status ttc_sen2cor_clean(ttc_ctx* c, const float* d_clm20, int32_t T, int32_t w20, int32_t h20, float* d_out, void* stream) {
    return c ? prep_sen2cor_clean(c, d_clm20, T, w20, h20, d_out, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_median5(ttc_ctx* c, const float* d_in, int32_t X, int32_t Y, float* d_out, void* stream) {
    return c ? prep_median5(c, d_in, X, Y, d_out, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_snow_map(ttc_ctx* c, const float* d_s2, int32_t T, int32_t X, int32_t Y, uint8_t* d_snow, int32_t* h_per_image, void* stream) {
    return c ? prep_snow(c, d_s2, T, X, Y, d_snow, h_per_image, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_merge_cloud_masks(ttc_ctx* c, float* d_cloudshad, float* d_clm, const uint8_t* d_fcps, int64_t n, void* stream) {
    return c ? prep_merge_clm(c, d_cloudshad, d_clm, d_fcps, n, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_count_positive(ttc_ctx* c, const float* d_a, int32_t T, int32_t npix, int32_t* h_counts, void* stream) {
    return c ? prep_count_positive(c, d_a, T, npix, NAN, h_counts, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_count_equal(ttc_ctx* c, const float* d_a, int32_t T, int32_t npix, float value, int32_t* h_counts, void* stream) {
    if (value != value) return c ? c->fail(TTC_ERR_ARG, "count_equal: value is NaN") : TTC_ERR_ARG;
    return c ? prep_count_positive(c, d_a, T, npix, value, h_counts, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_clip01(ttc_ctx* c, float* d_a, int64_t n, void* stream) {
    return c ? prep_clip01(c, d_a, n, TTC_S(stream)) : TTC_ERR_ARG;
}
ttc_status ttc_divide(ttc_ctx* c, float* d_a, int64_t n, float divisor, void* stream) {
    return c ? prep_divide(c, d_a, n, divisor, TTC_S(stream)) : TTC_ERR_ARG;
}
#undef TTC_S

ttc_status ttc_debug_knob(int32_t which, int32_t value) {
    static const bool enabled = [] { const char* e = getenv("TTC_ENABLE_PROBE_KNOBS"); return e && e[0] == '1'; }();
    if (!enabled) return TTC_ERR_STATE;          // process-wide, unsynchronised probe state: only for processes that asked for it
    h16_set_knob(which, value);
    return TTC_OK;
}

ttc_status ttc_debug_clouds_stage(ttc_ctx* c, int32_t stage) {
    if (!c) return TTC_ERR_ARG;
    c->clouds_debug_stage = stage;
    return TTC_OK;
}

ttc_status ttc_debug_keep(ttc_ctx* c, int32_t on) {
    if (!c) return TTC_ERR_ARG;
    c->keep_debug = on != 0;
    return TTC_OK;
}

ttc_status ttc_debug_fetch(ttc_ctx* c, const char* name, float* h_dst, size_t cap, size_t* n_floats) {
    if (!c || !name) return TTC_ERR_ARG;
    auto it = c->named.find(name);
    if (it == c->named.end()) return c->fail(TTC_ERR_ARG, std::string("unknown activation: ") + name);
    if (n_floats) *n_floats = it->second.second;
    if (!h_dst) return TTC_OK;
    if (!c->frames_planar_valid && std::string(name) == "frames")
        return c->fail(TTC_ERR_STATE, "frames: the last tile's windows were assembled in the 16-bit engine's blocked form only; pass d_model_in "
                                      "to ttc_predict_tile or switch ttc_debug_keep on to have the fp32 planar frames written");
    TTC_HIP(c, hipDeviceSynchronize());
    const size_t n = it->second.second < cap ? it->second.second : cap;
    TTC_HIP(c, hipMemcpy(h_dst, it->second.first, n * sizeof(float), hipMemcpyDeviceToHost));
    return TTC_OK;
}

ttc_status ttc_debug_timing(ttc_ctx* c, int32_t enable) {
    if (!c) return TTC_ERR_ARG;
    TTC_HIP(c, hipDeviceSynchronize());
    flush_timing(c);
    c->timing.level = enable;
    return TTC_OK;
}

ttc_status ttc_debug_kernel_ms(ttc_ctx* c, const char* name, double* avg_ms, int64_t* launches) {
    if (!c) return TTC_ERR_ARG;
    TTC_HIP(c, hipDeviceSynchronize());
    flush_timing(c);
    if (!name) { c->timing.recs.clear(); return TTC_OK; }
    auto it = c->timing.recs.find(name);
    if (it == c->timing.recs.end() || it->second.n == 0) {
        if (avg_ms) *avg_ms = 0.0;
        if (launches) *launches = 0;
        return TTC_OK;
    }
    if (avg_ms) *avg_ms = it->second.ms / (double)it->second.n;
    if (launches) *launches = it->second.n;
    return TTC_OK;
}

ttc_status ttc_debug_kernel_flops(ttc_ctx* c, const char* name, double* flops_per_launch, int64_t* launches) {
    if (!c || !name) return TTC_ERR_ARG;
    auto it = c->timing.recs.find(name);
    const bool have = it != c->timing.recs.end() && it->second.nf > 0;
    if (flops_per_launch) *flops_per_launch = have ? it->second.flops / (double)it->second.nf : 0.0;
    if (launches) *launches = have ? it->second.nf : 0;
    return TTC_OK;
}

}  // extern "C"
