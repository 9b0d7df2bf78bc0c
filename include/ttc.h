/*
 * ttc.h -- C ABI of libttc_hip.so: the MI355X (gfx950) implementation of the per-tile
 * numeric hot path of wri/sentinel-tree-cover.
 *
 * The reference has NO native interface for this path: it is pure Python that drives two
 * TensorFlow-1 frozen graphs through `Session.run` and numpy/scipy in between
 * (SURVEY.md F1/F2).  Each entry point below therefore cites the reference *Python*
 * function (file:line in the reference checkout) whose arithmetic it replaces; the
 * reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md (and shipped as sentinel-tree-cover_amd/_lib.py).
 *
 * Conventions
 *  - Every `d_*` pointer is a DEVICE pointer owned by the caller (e.g. a PyTorch-ROCm
 *    allocation); `h_*` pointers are host memory.  The library owns only its context
 *    (weights + workspace, sized at ttc_create).
 *  - All work is enqueued on the caller's `stream` (a hipStream_t passed as void*);
 *    no call synchronises the device except ttc_destroy, the ttc_debug_* helpers and where a
 *    function's comment says so (host read-backs).
 *  - Errors: integer status; ttc_last_error(ctx) returns a message.  No exceptions, no exit().
 *  - A context is bound to one device and is not re-entrant.
 *  - Array layouts are the reference's numpy layouts ([T, X, Y, C], C-contiguous float32)
 *    unless a parameter says "planar".
 *  - Non-finite inputs.  The model entry points expect FINITE window inputs (the reference feeds np.clip'ed, normalised values:
 *    job.py:316-325).  A NaN / Inf in one window makes that window's probabilities NaN -- in the reference too, its GroupNorm
 *    statistics span the whole window (model.py:100-121) -- and never touches another window of the batch; WHICH raw conv outputs
 *    inside the poisoned window are non-finite depends on `fp32_conv_form` (the Winograd forms transform 6 x 6 input patches).
 */
#ifndef TTC_H
#define TTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ttc_ctx ttc_ctx;

typedef enum {
    TTC_OK = 0,
    TTC_ERR_ARG = 1,      /* bad argument / unsupported geometry */
    TTC_ERR_HIP = 2,      /* a HIP runtime call failed */
    TTC_ERR_STATE = 3,    /* e.g. forward before weights are loaded */
    TTC_ERR_NOMEM = 4,
    TTC_ERR_IO = 5        /* a host file could not be opened / read / written */
} ttc_status;

/* Model / window geometry.  Defaults mirror src/download_and_predict_job.py:60-61, :1715
 * (SIZE = 172-14, length 4) and src/train/train-model.py:64-82 (base_filters 64, zoneout 0.75). */
typedef struct {
    int32_t win_in;        /* W: model input window, W % 4 == 0 (172, 168, ...)        */
    int32_t length;        /* L: ConvGRU steps (4 or 12); frames per window = L + 1     */
    int32_t max_windows;   /* workspace capacity in windows (36 per 618^2 tile)         */
    int32_t n_bands;       /* 17                                                        */
    int32_t hidden;        /* 32  (ConvGRU filters per direction = base_filters / 2)    */
    int32_t base_filters;  /* 64                                                        */
    float   zoneout;       /* 0.75: state' = z*state + (1-z)*new  (model.py:571-574)    */
    int32_t precision;     /* 0 = fp32 MFMA (exact fp32 FMA chains); 2 = fp16 / 3 = bf16 operands on the 16-bit engine:
                            * activations that feed a conv are stored as hi + lo 16-bit pairs, each layer multiplies 3
                            * split products (~fp32 accuracy for fp16 pairs) or 1 (plain 16-bit operands) per
                            * `one_term_layers`; accumulation, GroupNorm statistics and every non-conv operation stay
                            * fp32.  (1 and 4 were the bf16x3 / fp32-blocked engines, retired in round 3: rejected.)    */
    int32_t win_rows;      /* H: window rows for the non-square border graph of
                            * src/resegment_tiles_wide.py:478 ([L+1, SIZE_Y+14, SIZE+14, 17] = 220 x 684);
                            * 0 = square (win_in).  The per-tile core needs square windows. */
    uint32_t one_term_layers; /* precision 2 / 3 only: bit i set = conv layer i runs ONE 16-bit product (hi x hi) instead of
                            * three.  Bits: 0 ConvGRU gates, 1 ConvGRU candidate, 2..9 conv_median, conv_concat, conv1,
                            * conv2, up2, up2_out, up3, out, 10..15 the six DSen2 convs.  Default 0.  Measured on a real
                            * 618^2 tile with as-stored-scale kernels (tools/probes/tile_dprob_probe.py): ANY bit set moves
                            * max |dprob| to 3.0e-3 (bit 0) .. 2.7e-2, OUTSIDE the 1e-3 contract; all-zero: fp16 5e-5,
                            * bf16 3.7e-4.  Set bits only for workloads with a looser accuracy budget.                 */
    int32_t fp32_conv_form; /* precision 0 only: which form of the 3x3 convolution the GroupNorm layers run (the sums are the same
                            * fp32 sums; the forms differ in rounding, because Winograd's transforms cancel large intermediates):
                            * 0 = fastest: Winograd F(4x4,3x3) for the layers with a multiple of 64 output channels (ConvGRU gates
                            *     and the eight conv_swish_gn blocks), F(2x2,3x3) for the rest (ConvGRU candidate).  Raw conv
                            *     outputs up to 2e-4 from the fp64 oracle (values of magnitude 6), probabilities <= 5e-5;
                            * 1 = F(2x2,3x3) only (rounds 4's engine): raw outputs <= 5e-5, probabilities <= 3e-5;
                            * 2 = direct implicit GEMM only (9 multiply-accumulates per tap: the slowest, <= 2e-5).
                            * The contract of the path is 1e-3 on probabilities (BASELINE.json).                          */
    int32_t dsen2_precision; /* the DSen2 super-resolution convs (src/download_and_predict_job.py:95-147) only: 0 = the context's
                            * `precision`; 2 / 3 = run them on the 16-bit engine with fp16 / bf16 hi + lo operand pairs and all THREE
                            * split products (fp32 accumulate, fp32 bias / residual / tanh) even when `precision` is 0.  fp16 pairs
                            * carry 22 significant bits: <= 1e-5 on reflectance against the fp64 graph (fp32 engine: 6e-7), end to
                            * end on probabilities the same 3e-5 class as precision 0 (tests/test_gpu_tile.py, test_gpu_e2e.py);
                            * 2.4 ms instead of 5.0 ms per 618^2 x 12-date tile.  Ignored when `precision` is 2 / 3.  Default 0. */
    uint32_t two_term_layers; /* precision 2 (fp16 pairs) only: bit i set = conv layer i (bits 0..9 as in `one_term_layers`; the DSen2
                            * bits are ignored) multiplies TWO products, x_hi * (w_hi + w_lo): 16-bit ACTIVATIONS, exact weights -- the lo
                            * half of the layer's input is neither staged nor multiplied (2/3 of the matrix work, half the input bytes).
                            * `one_term_layers` wins where both bits are set.  Default 0.  Costed per layer on the CPU
                            * (profiles/r04_two_product_study.txt): gates 1.4e-4, candidate 5.0e-4, any U-Net block >= 1.6e-3 max |dprob|;
                            * measured for bits 0 | 1 (both ConvGRU convs): white-noise windows 1.7e-4 .. 7.0e-4 (tests/test_gpu_h16.py),
                            * but 3.9e-3 END TO END on bench.py's real tile (`alt_fp16_two_term`: 37.1 vs 35.3 Mpx/s) -- OUTSIDE the
                            * 1e-3 contract, like every one-product form.  Only for workloads with a looser accuracy budget.       */
} ttc_config;

/* A named host tensor in TensorFlow layout (conv kernels HWIO). */
typedef struct {
    const char*  name;
    const float* data;
    int32_t      ndim;
    int64_t      shape[4];
} ttc_tensor;

/* ---- lifecycle --------------------------------------------------------------------- */
const char* ttc_version(void);
/* sizeof(ttc_config) as THIS library was built: a binding (cgo / ctypes / JNI) compares it with its own struct before ttc_create -- the struct
 * grows at its end from round to round, and a stale binding would otherwise hand over a short struct silently */
size_t      ttc_config_size(void);
ttc_status  ttc_create(ttc_ctx** out, int32_t device, const ttc_config* cfg);
/* The same with the CALLER's sizeof(ttc_config): a binding built against an older header passes its shorter struct and the fields it does
 * not know read as zero (every field added after `win_rows` defaults to 0); a struct LONGER than this library's is refused with
 * TTC_ERR_ARG.  Bindings in other languages should call this one; ttc_create(cfg) == ttc_create_v2(cfg, ttc_config_size()). */
ttc_status  ttc_create_v2(ttc_ctx** out, int32_t device, const ttc_config* cfg, size_t cfg_size);
void        ttc_destroy(ttc_ctx* ctx);
const char* ttc_last_error(const ttc_ctx* ctx);
/* bytes of device memory the context holds (weights + workspace) */
size_t      ttc_device_bytes(const ttc_ctx* ctx);

/* ---- weights ------------------------------------------------------------------------
 * ConvGRU/U-Net variables (names: see sentinel-tree-cover_amd/weights.py; shapes as in
 * models-release/master-ckpt-nonfrozen/-0.meta).  Replaces
 * tf.import_graph_def of predict_graph-{W}.pb, src/download_and_predict_job.py:1800-1824. */
ttc_status ttc_load_weights(ttc_ctx* ctx, const ttc_tensor* tensors, int32_t n);
/* DSen2-lite variables ({in,01,02,11,12,out}_conv/{kernel,bias}).  Replaces the import of
 * models-release/supres-40k-swir/superresolve_graph.pb, job.py:1788-1796. */
ttc_status ttc_load_dsen2_weights(ttc_ctx* ctx, const ttc_tensor* tensors, int32_t n);

/* ---- model forward ------------------------------------------------------------------
 * == sess.run(predict_logits, {predict_inp: x, predict_length: L}) inside
 * predict_subtile, src/download_and_predict_job.py:353-357, batched over windows.
 * d_in : [n, L+1, H, W, 17] float32 (already normalised, job.py:316-325); H = win_rows or W
 * d_out: [n, H-14, W-14]    float32 probabilities                                      */
ttc_status ttc_forward_windows(ttc_ctx* ctx, const float* d_in, int32_t n, float* d_out, void* stream);

/* ---- precision calibration of the 16-bit engine ----------------------------------------------------------------------------------
 * Which conv layers of a 16-bit context (precision 2 / 3) need all three split products is a property of the WEIGHTS (GroupNorm,
 * src/train/src/model.py:100-121, divides by the per-group deviation of the conv output and amplifies a 2^-11 operand error where that
 * deviation is small) and of how smooth the inputs are.  This call measures it for the caller's model and data instead of assuming it:
 * d_windows [n, L+1, H, W, 17] -- normalised model inputs of REAL windows, e.g. the d_model_in of ttc_predict_tile (white-noise windows
 * under-state the error of real, spatially smooth tiles by ~8 x: DESIGN.md 4.1c) -- go through `ref_ctx` (an fp32 context of the same
 * geometry on the same device with the same weights loaded) and through candidate maps on `ctx`; layers are visited by matrix work,
 * largest first, and each takes ONE product (x_hi * w_hi), else TWO (x_hi * (w_hi + w_lo), fp16 only), where the probabilities of the
 * WHOLE map stay within `budget` of the fp32 engine's.  The chosen map is left applied to `ctx` (as if it had been created with these
 * `one_term_layers` / `two_term_layers`; the DSen2 bits keep their configured value) and reported.  BASELINE's contract is 1e-3 on
 * probabilities: a budget of 5e-4 leaves half of it to the rest of the chain.  Synchronises the stream (about 45 forwards of n windows). */
#define TTC_CAL_LAYERS 10    /* bits 0..9 of one_term_layers: ConvGRU gates, candidate, conv_median, conv_concat, conv1, conv2, up2, up2_out, up3, out */
typedef struct {
    uint32_t one_term_layers, two_term_layers;   /* the map applied */
    float    max_dprob;                          /* max |p(map) - p(fp32)| over the sample windows                                  */
    float    dprob_all_three;                    /* the same with three products everywhere: the floor no map can get under         */
    float    layer_dprob_one[TTC_CAL_LAYERS];    /* layer l ALONE on one product, the others on three                               */
    float    layer_dprob_two[TTC_CAL_LAYERS];    /* ... on two products (NaN for bf16 contexts)                                     */
    float    budget;
    int32_t  within_budget;                      /* 0: even three products everywhere exceed the budget on these windows            */
    double   matrix_work_ratio;                  /* matrix work of the map / matrix work with three products everywhere (1/3 .. 1)  */
    int32_t  trials, n_windows;
} ttc_precision_report;
ttc_status ttc_calibrate_precision(ttc_ctx* ctx, ttc_ctx* ref_ctx, const float* d_windows, int32_t n, float budget,
                                   ttc_precision_report* report, void* stream);

/* Forward + the two feature tensors of --gen_feats (job.py:1429-1445; tensors named at :1808-1809):
 * d_early [n, H, W, 64]       == sess.run("predict/gru_drop/drop_block2d/cond/Merge:0")  (bi-ConvGRU output)
 * d_late  [n, H-14, W-14, 64] == sess.run("predict/csse_out_mul/mul:0")                  (last block after its sSE gate)
 * either may be NULL.  With d_late the eight U-Net blocks of THIS forward run in the F(2x2) form whatever `fp32_conv_form` says (the late
 * features leave as int16 thousandths, job.py:174-180: they keep round 4's <= 5e-5-class raw-output accuracy); d_out comes from the same forward. */
ttc_status ttc_forward_taps(ttc_ctx* ctx, const float* d_in, int32_t n, float* d_out, float* d_early, float* d_late,
                            void* stream);

/* ---- per-tile numeric core ----------------------------------------------------------
 * == process_subtiles (job.py:1125-1483) up to and including the per-window post-masks,
 * for one tile whose dates have already been screened on the host (deal_w_missing_px's
 * date removal, job.py:1031-1037, is a host decision taken from ttc_tile_missing_counts).
 *
 * d_s2     [T, X, Y, 10]  cloud-free Sentinel-2 (output of process_tile + superresolve)
 * h_wmat   [12, T]        float32 host: temporal operator = Whittaker(lambda=100, 24->12)
 *                         o date-regrid (src/preprocessing/whittaker_smoother.py:25-67,
 *                         src/downloading/utils.py:176-347), built by the host mirror
 * h_keep   [T] int32 host  1 = date survives deal_w_missing_px's screening (job.py:1032-1037);
 *                         NULL = all.  Medians (job.py:1152-1160) use ALL dates, smoothing and
 *                         the clear-image count only the kept ones, as in the reference.
 * d_interp [T, X, Y]      interpolated-area weights from the gap-fill
 * d_s1     [12, X, Y, 2]  monthly Sentinel-1 (dB-scaled)
 * d_dem    [X, Y]
 * h_min/h_max [17] double normalisation vectors (job.py:1829-1842; Python floats in the reference:
 *                         midrange / half-range are formed in float64, then used as float32)
 * size                    output window size (W - 14); windows follow job.py:1295-1317
 * n_dates_ok              len(dates) after host screening (job.py:1418: < 2 -> no data)
 * d_windows [36, size, size] float32: what the reference np.save()s per window
 *                         (rounded to 3 decimals, 255 = no data), order = window index
 * d_windows_raw           same before np.around / bright-surface product (may be NULL)
 */
ttc_status ttc_process_subtiles(ttc_ctx* ctx, const float* d_s2, int32_t T, int32_t X, int32_t Y,
                                const float* h_wmat, const int32_t* h_keep, const float* d_interp,
                                const float* d_s1, const float* d_dem, const double* h_min,
                                const double* h_max, int32_t size, int32_t n_dates_ok,
                                float* d_windows, float* d_windows_raw, void* stream);

/* per-date count of "missing" pixels, id_missing_px (src/preprocessing/interpolation.py:5-23);
 * d_counts [T] int32 */
ttc_status ttc_tile_missing_counts(ttc_ctx* ctx, const float* d_s2, int32_t T, int32_t X, int32_t Y,
                                   int32_t* d_counts, void* stream);
/* in-place repair of NaN / 0 / 1 samples with the running temporal median:
 * interpolate_na_vals (interpolation.py:42-56) then deal_w_missing_px's value fixes
 * (job.py:1039-1047).  d_s2 [T, X, Y, 10]. */
ttc_status ttc_tile_fix_missing(ttc_ctx* ctx, float* d_s2, int32_t T, int32_t X, int32_t Y,
                                int32_t do_nan, int32_t do_zero_one, void* stream);

/* ---- cloud / shadow gap-fill (src/preprocessing/cloud_removal.py) --------------------
 * Feather weights == id_areas_to_interp (cloud_removal.py:774-798; closing 15, clip 1) or the
 * identical block inside remove_cloud_and_shadows (:910-923; closing 20, clip 0).
 * d_mask [T, X, Y] float32 (binary cloud+shadow mask) -> d_w [T, X, Y] float32 in [0, 1]. */
ttc_status ttc_feather(ttc_ctx* ctx, const float* d_mask, int32_t T, int32_t X, int32_t Y, int32_t closing,
                       int32_t clip, float* d_w, void* stream);
/* == make_aligned_mosaic (cloud_removal.py:578-699).  d_tiles [T, X, Y, 10]; d_w [T, X, Y] in/out
 * (dates that cannot be aligned are set to 1, :679-680); d_mosaic [X, Y, 10] out.  Row counts, selection ranks
 * and the per-date statistics are computed on the device for all dates at once; the call waits for the stream
 * ONCE to read T flags back (does any date have <= 1000 usable rows?) and then either finishes in one launch or
 * falls back to the reference's date-by-date order. */
ttc_status ttc_aligned_mosaic(ttc_ctx* ctx, const float* d_tiles, float* d_w, int32_t T, int32_t X, int32_t Y,
                              float* d_mosaic, void* stream);
/* Row sampler for the per-date fit (align_interp_array_randomforest, cloud_removal.py:453-505):
 * given the unclipped EVI of the n_rows candidate training rows (host memory), write the chosen row
 * indices (repeats allowed) to out_idx (capacity cap) and return their number.  The Python mirror
 * replays the reference's stdlib random.shuffle sequence through this hook. */
typedef int64_t (*ttc_sampler_fn)(const float* h_evi, int64_t n_rows, int64_t* out_idx, int64_t cap, void* user);
/* == remove_cloud_and_shadows (cloud_removal.py:888-973): feather (closing 20) -> aligned mosaic ->
 * per date: non-negative least-squares map [mosaic(10), snow] -> band (10 fits), predict, blend ->
 * residual clouds in the mosaic (calculate_clouds_in_mosaic, :703-732).
 * d_tiles [T, X, Y, 10] is modified in place; d_probs [T, X, Y] binary mask; d_pfcps [X, Y] uint8 or NULL;
 * sampler NULL = deterministic expected-multiplicity weighting on the device (no RNG);
 * d_interp [T, X, Y] out (areas_interpolated); d_mosaic [X, Y, 10] out or NULL;
 * h_to_remove [T] / n_to_remove: dates that are interpolated everywhere (:958-959); filling them is the
 * second place this call waits for the stream (the first is the aligned mosaic's flag read-back).
 * With a `sampler` the call additionally synchronises once per date to hand the EVI column to the host. */
ttc_status ttc_remove_cloud_and_shadows(ttc_ctx* ctx, float* d_tiles, const float* d_probs, const uint8_t* d_pfcps,
                                        int32_t T, int32_t X, int32_t Y, ttc_sampler_fn sampler, void* user,
                                        float* d_interp, float* d_mosaic, int32_t* h_to_remove,
                                        int32_t* n_to_remove, void* stream);

/* ---- cloud / shadow DETECTION (the step before the gap-fill) -----------------------
 * == identify_clouds_shadows(img, dem, bbx) incl. detect_pfcp (cloud_removal.py:1215-1677, :1109-1212).
 * d_img [T, X, Y, 10] float32 (10 m stack after the bilinear upsample); d_dem [X, Y] metres.
 * The two ESA-WorldCover rasters the reference opens with rasterio are passed as full-resolution masks:
 *   d_forest [X, Y] uint8      == adjust_cloudmask_in_forests(...)             NULL = none (the reference's except branch)
 *   d_urban_core / d_urban_near [X, Y] uint8 == the two resized masks of mask_nonurban_areas (:735-755); both NULL = none
 * d_clouds [T, X, Y] float32 in {0, 1} (clouds + shadows), d_fcps [T, X, Y] uint8 (potential false-positive mask).
 * X and Y must be even when the urban masks are given (2x2 parallax grid).  Waits for the stream once (tiny table upload). */
ttc_status ttc_identify_clouds_shadows(ttc_ctx* ctx, const float* d_img, int32_t T, int32_t X, int32_t Y, const float* d_dem,
                                       const uint8_t* d_forest, const uint8_t* d_urban_core, const uint8_t* d_urban_near,
                                       float* d_clouds, uint8_t* d_fcps, void* stream);

/* ---- small raster steps of process_tile between the stages above (job.py:641-995) ------------------
 * Sen2Cor mask clean-up (:688-697): d_clm20 [T, w20, h20] float32 0/1 -> d_out [T, 2*w20, 2*h20]; walking the dates in order,
 * two consecutive flagged dates at a pixel are both cleared. */
ttc_status ttc_sen2cor_clean(ttc_ctx* ctx, const float* d_clm20, int32_t T, int32_t w20, int32_t h20, float* d_out, void* stream);
/* dem = median_filter(dem, size = 5) (:713; scipy 'reflect' border); not in place. */
ttc_status ttc_median5(ttc_ctx* ctx, const float* d_in, int32_t X, int32_t Y, float* d_out, void* stream);
/* snow map (:799-821): d_snow [X, Y] uint8 = 1 - dilate(mean_t(snow_filter) < 0.7, 2); h_per_image [T] (may be NULL) receives
 * the number of snow pixels per date (the > 25 % rule of :822 is a host decision) -- filling it waits for the stream. */
ttc_status ttc_snow_map(ttc_ctx* ctx, const float* d_s2, int32_t T, int32_t X, int32_t Y, uint8_t* d_snow, int32_t* h_per_image, void* stream);
/* cloudshad = max(cloudshad, clm) after clm[fcps] = 0 when d_fcps is given (:843-848 / :880-884); n elements, in place. */
ttc_status ttc_merge_cloud_masks(ttc_ctx* ctx, float* d_cloudshad, float* d_clm, const uint8_t* d_fcps, int64_t n, void* stream);
/* h_counts[t] = #(d_a[t] > 0) (np.mean(interp > 0, axis = (1, 2)) of :868 is count / npix); waits for the stream. */
ttc_status ttc_count_positive(ttc_ctx* ctx, const float* d_a, int32_t T, int32_t npix, int32_t* h_counts, void* stream);
/* per date, how many pixels equal `value` (np.mean(interp == 1, axis = (1, 2)) of resegment_tiles_wide.py:651) */
ttc_status ttc_count_equal(ttc_ctx* ctx, const float* d_a, int32_t T, int32_t npix, float value, int32_t* h_counts, void* stream);
/* np.clip(x, 0, 1) in place (:994). */
ttc_status ttc_clip01(ttc_ctx* ctx, float* d_a, int64_t n, void* stream);
/* x / divisor in place with an IEEE division (dem / 90, :993). */
ttc_status ttc_divide(ttc_ctx* ctx, float* d_a, int64_t n, float divisor, void* stream);

/* ---- whole tile in ONE call -------------------------------------------------------------------------------------------------
 * The chain the job runs per tile (job.py:1995-2020) for a tile on which none of process_tile's date-DROPPING rules fires,
 * enqueued on `stream` WITHOUT any host round trip:
 *   to_float32 (tof_downloading.py:64-72), convert_to_db (job.py:74-89 / :699-708), the 20 m -> 10 m bilinear (:734-782),
 *   [identify_clouds_shadows (cloud_removal.py:1215-1677) with TTC_TILE_DETECT, else the GIVEN mask],
 *   remove_cloud_and_shadows (:888-973) with the deterministic expected-multiplicity sampler, the final np.clip(., 0, 1)
 *   (job.py:993), superresolve_large_tile (:95-147), process_subtiles (:1125-1483; the date screening of deal_w_missing_px
 *   :1031-1037 and the 12 x T temporal operator are formed on the device), load_mosaic_predictions (:1515-1641).
 * NOT in the call: the DEM median filter / division (d_dem is passed as process_tile returns it), the Sen2Cor mask (merge
 * it into d_mask beforehand; with TTC_TILE_DETECT a tile that has one goes through the staged calls), the snow map
 * (an output of process_tile nothing downstream reads), and every rule of process_tile that REMOVES dates -- those change
 * T and re-run the detection, so the call evaluates their conditions on the device, carries on speculatively with all
 * dates, and REPORTS them.  The caller reads d_status after synchronising the stream; if d_status[0], [2] or [3] is
 * non-zero the rasters are NOT the reference's and the tile must be re-run through the staged calls
 * (job.predict_tile_raw_checked does exactly that):
 *   d_status[0] != 0  a date could not be radiometrically aligned (cloud_removal.py:679-680: it marks itself fully
 *                     interpolated, which changes every later date; ttc_remove_cloud_and_shadows takes that branch itself)
 *   d_status[1]       dates that survived the missing-pixel screening (< 2 -> every window is 255, like job.py:1418-1422);
 *                     informational
 *   d_status[2] != 0  number of dates the gap-fill flagged as fully interpolated; the job deletes them before
 *                     process_subtiles (job.py:964-981)
 *   d_status[3]       bit mask of process_tile's other date-dropping rules: 1 = a date has >= X^2 / 2 missing pixels
 *                     (id_missing_px(., 2), job.py:786); 2 = more than 10 dates are > 25 % snow (:822); 4 = the feathered mask
 *                     of a date covers > 90 % of the tile (:866-921; evaluated on the closing-20 weights, which bound the
 *                     closing-15 weights of id_areas_to_interp from above: conservative, never missed)
 * d_s2_10 [T, X, Y, 4] / d_s2_20 [T, X/2, Y/2, 6] / d_s1 [12, X, Y, 2] uint16 as stored in temp/raw (device memory);
 * d_dem [X, Y] as process_tile returns it (median-filtered, / 90); d_dem_m the same in metres (detection only, may be NULL);
 * d_mask [T, X, Y] the cloud + shadow mask (ignored with TTC_TILE_DETECT); d_dates [T] int32 day of year, DEVICE memory.
 * d_out_u8 [Y, X] uint8 (transposed like the reference's mosaic, job.py:1578; 255 = no data), d_out_f32 the float percent raster
 * of the same shape or NULL;
 * d_model_in (optional) receives the model's input frames [36, L+1, 17, W+2, W+2] (planar, padded: what ttc_debug_fetch
 * "frames" returns).  ttc_debug_fetch "pt_windows" / "pt_windows_raw" [36, size, size]: the per-window arrays the reference
 * np.save()s, and the same before np.around / the bright-surface product.  flags: TTC_TILE_*. */
#define TTC_TILE_DETECT 1        /* run the multi-temporal cloud / shadow detection and gap-fill with its mask           */
#define TTC_TILE_INPUTS_ONLY 2   /* stop after the model inputs are assembled: preprocessing only (BASELINE configs[2])   */
#define TTC_TILE_NO_SUPERRES 4   /* skip the DSen2 super-resolution                                                     */
ttc_status ttc_predict_tile(ttc_ctx* ctx, const uint16_t* d_s2_10, const uint16_t* d_s2_20, const uint16_t* d_s1,
                            const float* d_dem, const float* d_dem_m, const float* d_mask, const int32_t* d_dates,
                            int32_t T, int32_t X, int32_t Y, const double* h_min, const double* h_max, int32_t size,
                            int32_t flags, uint8_t* d_out_u8, float* d_out_f32, float* d_model_in, int32_t* d_status,
                            void* stream);

/* The same call for raw arrays that are NOT all on one grid -- the case adjust_shape exists for.  process_tile keys the tile's size on the
 * 20 m stack (width, height = 2 x s2_20.shape[1:3], job.py:716-717) and crops / edge-pads Sentinel-1 (after its dB conversion, :699-718), the
 * 10 m bands (:720) and the median-filtered DEM (:713, :721) to it with adjust_shape (:260-310): an array one pixel short gains its first row /
 * column once more, one pixel long loses it, an even difference is split between both ends.  Here that is an index map inside the decode
 * passes (no copy of the big arrays); only the DEM planes are re-gridded into scratch.  Every array comes with ITS OWN rows x cols:
 *   shapes->s2_20  [h, w]     X = 2 h, Y = 2 w are the tile's grid and the shape of every output (d_out_u8 [Y, X] ...)
 *   shapes->s2_10, ->s1, ->dem   as stored; d_dem / d_dem_m share ->dem
 *   shapes->mask   must equal [X, Y] (the mask is produced on the tile's grid, :839; ignored with TTC_TILE_DETECT)
 * Differences adjust_shape does not reconcile (an odd difference of 3 or more: the reference's own result has the wrong length there and
 * process_tile raises on the next statement) and a mask of another shape return TTC_ERR_ARG with the offending array named in
 * ttc_last_error.  Everything else as ttc_predict_tile, which is this call with every shape equal to [X, Y]. */
typedef struct {
    int32_t s2_10[2], s2_20[2], s1[2], dem[2], mask[2];
} ttc_tile_shapes;
ttc_status ttc_predict_tile_shaped(ttc_ctx* ctx, const uint16_t* d_s2_10, const uint16_t* d_s2_20, const uint16_t* d_s1,
                                   const float* d_dem, const float* d_dem_m, const float* d_mask, const int32_t* d_dates, int32_t T,
                                   const ttc_tile_shapes* shapes, const double* h_min, const double* h_max, int32_t size,
                                   int32_t flags, uint8_t* d_out_u8, float* d_out_f32, float* d_model_in, int32_t* d_status,
                                   void* stream);
/* adjust_shape (job.py:260-310) on its own: d_in [T, n1, n2, channels] float32 -> d_out [T, width, height, channels] (not in place).
 * TTC_ERR_ARG where the reference's rule does not produce width x height (see above). */
ttc_status ttc_adjust_shape(ttc_ctx* ctx, const float* d_in, int32_t T, int32_t n1, int32_t n2, int32_t channels, int32_t width,
                            int32_t height, float* d_out, void* stream);

/* ---- Gaussian overlap mosaic --------------------------------------------------------
 * == load_mosaic_predictions(out_folder, depth=1), job.py:1515-1641, from the 36 window
 * arrays (not from .npy files).
 * d_windows [n, size, size]; h_xy [n, 2] int32 = (folder_x, folder_y) of each window.
 * d_out_u8 [max_y+size, max_x+size] uint8 -- TRANSPOSED like the reference (job.py:1578);
 * d_out_f32 same shape, float32 percent before quantisation, NaN = no data (may be NULL). */
ttc_status ttc_mosaic(ttc_ctx* ctx, const float* d_windows, int32_t n, const int32_t* h_xy,
                      int32_t size, int32_t out_rows, int32_t out_cols,
                      uint8_t* d_out_u8, float* d_out_f32, void* stream);

/* == load_mosaic_predictions(feats_folder, depth = 64), job.py:1552-1592 (feature export): Gaussian blend of the
 * int16 feature windows.  d_feats [n, size, size, depth] int16; h_xy as above; d_out [depth, out_rows, out_cols] int16,
 * 0 where no window covers a pixel. */
ttc_status ttc_mosaic_features(ttc_ctx* ctx, const int16_t* d_feats, int32_t n, const int32_t* h_xy, int32_t size,
                               int32_t depth, int32_t out_rows, int32_t out_cols, int16_t* d_out, void* stream);

/* ---- tile-border resegmentation ---------------------------
 * == process_subtiles of src/resegment_tiles_wide.py:360-616 for one border strip, up to the window predictions that
 * the reference np.save()s: NaN fix, median / quarterly medians of the 12 steps, per-window histogram alignment of
 * the two halves (align_subtile_histograms, :284-343), 7-row reflect pad of the first / last window, 17-channel
 * assembly + float32 normalisation (:199-200), forward with the non-square graph, seam adjustment (:518-531).
 * The context must have win_in = SIZE+14 (strip width), win_rows = SIZE_Y+14, length 4.
 *
 * d_s2   [12, X, W, 14]  smoothed + super-resolved bands (10) and smoothed indices (4) of the strip
 * d_s1   [12, X, W, 2]   d_dem [X, W]
 * h_rows [n, 4] int32    per window: first strip row, rows taken, reflect-pad rows before / after
 *                        (tiles_array of make_tiles_right_neighb, :267-281, with the padding rule of :463-472)
 * h_min / h_max [17]     float32 normalisation vectors (:1664-1685)
 * hist_align             != 0: align the halves (the reference's hist_align flag, :1140-1145)
 * n_dates_ok             len(dates); < 2 -> every window is the 255 fill (:505-508)
 * d_preds [n, H-14, W-14] what the reference saves per window (all-zero window -> 255 fill, :196-220)
 * h_stats [n, 4] float32 host: max, mean of the window prediction, 1 if the seam adjustment fired, 1 if 255-filled
 *                        (inputs of the keep / skip rule, :534-613, which stays on the host)
 * h_applied [n, 5] int32 host, may be NULL: 1 where the alignment of frame f (4 = median frame) was kept
 * Synchronises the stream before returning. */
ttc_status ttc_border_subtiles(ttc_ctx* ctx, const float* d_s2, const float* d_s1, const float* d_dem, int32_t X,
                               const int32_t* h_rows, int32_t n, const float* h_min, const float* h_max, int32_t hist_align,
                               int32_t n_dates_ok, float* d_preds, float* h_stats, int32_t* h_applied, void* stream);

/* regularize_and_smooth (resegment_tiles_wide.py:772-790) + make_and_smooth_indices (job.py:1009-1028) for a border strip:
 * d_s2 [T, X, Y, 10] (gap-filled, deal_w_missing_px applied) -> d_out [12, X, Y, 14] = 12 monthly steps of the 10 bands and
 * of evi / bi / msavi2 / grndvi computed per date; h_wmat [12, T] as for ttc_process_subtiles. */
ttc_status ttc_smooth_strip(ttc_ctx* ctx, const float* d_s2, int32_t T, int32_t X, int32_t Y, const float* h_wmat, float* d_out,
                            void* stream);

/* superresolve_large_tile with the window edge and pixel stride as parameters: wsize = 125 and channels = 14 reproduce
 * resegment_tiles_wide.py:144-179 on the array above (bands 4..9 of every pixel are replaced; channels >= 10 untouched);
 * wsize = 110, channels = 10 is ttc_superresolve_tile.  quirks as there. */
ttc_status ttc_superresolve_windows(ttc_ctx* ctx, float* d_arr, int32_t T, int32_t X, int32_t Y, int32_t channels, int32_t wsize,
                                    int32_t quirks, void* stream);

/* The seam adjustment of ttc_border_subtiles on its own (resegment_tiles_wide.py:518-531), in place on
 * d_preds [n, rows, cols]: when the means of the 4 columns either side of cols/2 differ by more than 0.15, the values
 * > 0.05 of each half move by half the difference of the halves' means (over values > 0.05), then clip to [0, 1].
 * h_stats [n, 4] as above.  Synchronises the stream. */
ttc_status ttc_seam_adjust(ttc_ctx* ctx, float* d_preds, int32_t n, int32_t rows, int32_t cols, float* h_stats, void* stream);

/* One saved window of a tile folder, as recreate_resegmented_tifs (resegment_tiles_wide.py:1240-1549) finds it. */
typedef struct {
    int32_t kind;        /* 0 `{x}/{y}.npy`, 1 `{x}/left{y}.npy`, 2 `right{x}/{y}.npy`, 3 `{x}/up{y}.npy`, 4 `{x}/down{y}.npy` */
    int32_t x, y;        /* the x / y encoded in the path */
    int32_t rows, cols;  /* shape of the saved array */
    int64_t pred_off;    /* offset (floats) of the array in d_preds */
    int64_t weight_off;  /* offset (floats) in d_weights of its weight table [sx, sy]: fspecial_gauss (kind 0, :1303-1318) or
                          * the resized half Gaussian (:1338-1349 and siblings); sx, sy = the part kept (half for kinds 1-4) */
} ttc_reseg_window;

/* == recreate_resegmented_tifs + mosaic_subtiles (:1169-1549) with the directory walk replaced by a table.
 * d_ramps [5, X, Y] float64: the stack-vs-stack weights `m` of mosaic_subtiles for kinds n, l, r, u, d (planes of
 * absent kinds are not read); X = shape[1], Y = shape[0].  d_out [X, Y] float32 0..100, 255 = no data; d_sums [X, Y] may be
 * NULL.  Plain windows that do not fit the tile are dropped by the caller (:1315). */
ttc_status ttc_reseg_mosaic(ttc_ctx* ctx, const float* d_preds, const ttc_reseg_window* h_wins, int32_t n,
                            const float* d_weights, const double* d_ramps, int32_t X, int32_t Y, float* d_out, float* d_sums,
                            void* stream);

/* ---- 20 m -> 10 m -------------------------------------------------------------------
 * DSen2-lite on one padded window batch: == sess.run(superresolve_logits, ...) in
 * superresolve_large_tile._worker_fn, job.py:112-118.
 * d_in [n, H, W, 10], d_bilinear [n, H, W, 6] -> d_out [n, H, W, 6] */
ttc_status ttc_dsen2_forward(ttc_ctx* ctx, const float* d_in, const float* d_bilinear, int32_t n,
                             int32_t H, int32_t W, float* d_out, void* stream);
/* whole-tile driver == superresolve_large_tile (job.py:95-147), in place on d_s2 [T, X, Y, 10];
 * quirks != 0 reproduces the reference's skipped strip and double pass. */
ttc_status ttc_superresolve_tile(ttc_ctx* ctx, float* d_s2, int32_t T, int32_t X, int32_t Y,
                                 int32_t quirks, void* stream);
/* bilinear x2 of the 20 m bands == the resize() loop of process_tile, job.py:734-782 (even grids):
 * d_s2_10 [T, 2h, 2w, 4], d_s2_20 [T, h, w, 6] -> d_out [T, 2h, 2w, 10] */
ttc_status ttc_upsample_20m(ttc_ctx* ctx, const float* d_s2_10, const float* d_s2_20, int32_t T,
                            int32_t h, int32_t w, float* d_out, void* stream);

/* ---- storage codecs / Sentinel-1 scaling ------------------------------------------------
 * to_float32 (src/tof/tof_downloading.py:64-72): uint16 / 65535 -> float32;  to_int16 (:51-61):
 * trunc(clip(x, 0, 1) * 65535) -> uint16.  n = element count. */
ttc_status ttc_u16_to_float(ttc_ctx* ctx, const uint16_t* d_in, int64_t n, float* d_out, void* stream);
ttc_status ttc_float_to_u16(ttc_ctx* ctx, const float* d_in, int64_t n, uint16_t* d_out, void* stream);
/* float_to_int16 (job.py:174-180): NaN -> -32768; trunc(clip(x, -32768/precision, 32767/precision) * precision). */
ttc_status ttc_float_to_int16(ttc_ctx* ctx, const float* d_in, int64_t n, float precision, int16_t* d_out, void* stream);
/* Sentinel-1 preparation of process_tile (job.py:699-708): /65535, saturated (== 1) samples -> the image's
 * median, convert_to_db(., 22) (job.py:74-89).  d_u16 [T, X, Y, 2] -> d_out [T, X, Y, 2] float32. */
ttc_status ttc_s1_to_db(ttc_ctx* ctx, const uint16_t* d_u16, int32_t T, int32_t X, int32_t Y, float* d_out, void* stream);

/* ---- on-disk formats (host code, no GPU) ----------------------------------------------------------------------------------
 * hkl.load(path) for the numeric arrays of temp/raw (src/download_and_predict_job.py:684-714; written at :462-463, :592-633
 * with hkl.dump(..., compression='gzip')): reads the dataset at `name`, a '/'-separated path from the root ("data",
 * "data/data_1"); NULL = what hkl.load returns first: hickle 4/5's "data", then hickle 3's "data_0", then the first dataset,
 * descending into container groups (a dumped list of arrays is a group of data_i datasets) -- into h_out (raw little-endian
 * elements, C order).  Supported: the layout h5py's default format produces -- superblock v0/v1, old-style groups, object
 * headers v1 (attributes skipped), contiguous or chunked (B-tree v1) storage, deflate and shuffle filters.  h_out may be NULL to
 * query shape[<= 8] / ndim / elem_size / type_class (0 integer, 1 float) / is_signed only.  ttc_read_hkl_error() returns the
 * message of the last failure.  Pinned against files written by h5py 3.3.0 / HDF5 1.10.6 in hickle's layouts
 * (tests/golden/hkl, tools/gen_golden_hkl.py); hickle itself is not installed in the build image. */
ttc_status ttc_read_hkl(const char* path, const char* name, void* h_out, size_t cap_bytes, int64_t* shape, int32_t* ndim,
                        int32_t* elem_size, int32_t* type_class, int32_t* is_signed);
const char* ttc_read_hkl_error(void);

/* ---- output file (host-side) -------------------------------
 * == write_tif (src/downloading/io.py:229-263) without rasterio: h_raster [rows, cols] uint8 host memory, already in the
 * file's orientation (write_tif transposes `arr`; load_mosaic_predictions' [Y, X] raster goes in after `.T` like there),
 * bounds = point[0], point[1], point[2], point[3] (west, south, east, north).  Classic TIFF, LZW strips,
 * ModelPixelScale / ModelTiepoint as rasterio.transform.from_bounds gives them, EPSG:4326 GeoKeys.  No context needed. */
ttc_status ttc_write_geotiff_u8(const char* path, const uint8_t* h_raster, int32_t rows, int32_t cols, double west, double south,
                                double east, double north);

/* ---- introspection for parity tests -------------------------------------------------
 * Copies a named internal activation (device) to host after synchronising the device.
 * Returns TTC_ERR_ARG for unknown names; *n_floats is the element count.  Test aid only. */
/* on != 0: subsequent forwards also write intermediates that the fused kernels otherwise keep in registers ("u"). */
ttc_status ttc_debug_keep(ttc_ctx* ctx, int32_t on);
ttc_status ttc_debug_fetch(ttc_ctx* ctx, const char* name, float* h_dst, size_t cap_floats,
                           size_t* n_floats);
/* average device time (ms) of the named kernel family over the launches since the last
 * reset, measured with HIP events on the launch stream; name == NULL resets.  Collected after
 * ttc_debug_timing(ctx, level): 0 = off, 1 = every kernel family, 2 = conv-engine launches only
 * (cheap enough to leave on inside a timed benchmark region). */
ttc_status ttc_debug_timing(ttc_ctx* ctx, int32_t enable);
ttc_status ttc_debug_kernel_ms(ttc_ctx* ctx, const char* name, double* avg_ms, int64_t* launches);
/* conv-engine families ("conv_gates", "conv_cand", the U-Net block names, "dsen2_conv") only: the flops of the MATRIX INSTRUCTIONS the
 * average launch since the last reset issues -- workgroup tiles x k-steps x flops per MFMA with the padding the kernel really multiplies;
 * the Winograd forms issue 2.25 / 4 multiply-accumulates where the direct form issues 9.  flops / avg_ms = the matrix pipe's rate, the
 * number bench.py prices against the MFMA peak.  Collected under the same ttc_debug_timing levels. */
ttc_status ttc_debug_kernel_flops(ttc_ctx* ctx, const char* name, double* flops_per_launch, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* TTC_H */
