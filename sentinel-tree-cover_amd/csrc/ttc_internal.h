// Internal declarations shared by the HIP translation units of libttc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ttc.h"

// ---- probe / bisecting entry points: exported (tools/probes/*.py and the cloud-detector stage tests bind them through ctypes) but NOT
// part of the drop-in surface, hence declared here and not in include/ttc.h (sentinel-tree-cover_amd/_lib.py: INTERNAL_EXPORTS)
extern "C" {
/* stage != 0: ttc_identify_clouds_shadows returns the flag planes after that stage of the detector (bisecting aid). */
ttc_status ttc_debug_clouds_stage(ttc_ctx* ctx, int32_t stage);
/* Device-side memory check without a sanitizer (GPU AddressSanitizer / xnack are not available on the target pool): with TTC_GUARD=<KiB> in the
 * environment every device buffer the library owns (weights, workspace, scratch) is allocated with a guard zone of that size before and after
 * it, filled with 0xA5.  This call scans every guard zone on the device and reports the bytes that no longer hold the pattern (an out-of-bounds
 * WRITE of some kernel) and the first offending buffer in msg ("<name> +<offset> after|before").  *n_bad = 0 and TTC_OK = clean;
 * TTC_ERR_STATE when the library runs without guards.  Synchronises the device. */
ttc_status ttc_debug_check_guards(ttc_ctx* ctx, int64_t* n_bad, char* msg, size_t cap);
/* PROBE ONLY -- not part of the drop-in surface.  Process-wide knobs of the 16-bit conv engine (tools/probes/h16_knobs.py, h16_trace.py):
 * which 0 = persistent grid size (-1 default = 2 per CU, 0 = one workgroup per tile), 1 = start offset of the odd wave slot in
 * s_sleep(127) units (-1 default), 2 | 3 = halves of a device pointer to a trace buffer, 4 = epilogue kind to trace.  They are plain
 * process globals read at launch time by every context and stream, so the call is REFUSED (TTC_ERR_STATE) unless the process was
 * started with TTC_ENABLE_PROBE_KNOBS=1 in its environment. */
ttc_status ttc_debug_knob(int32_t which, int32_t value);
}

#define TTC_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->fail(TTC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
            return TTC_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define TTC_CHECK(expr)                         \
    do {                                        \
        ttc_status s_ = (expr);                 \
        if (s_ != TTC_OK) return s_;            \
    } while (0)

// ----------------------------------------------------------------------------- conv engine
// Implicit-GEMM 3x3 convolution on fp32 MFMA over "flattened padded planes":
//   in  : [n][Cin][Hp*Wp]  planar, already padded (reflect / zero / none) by the producer
//   out : out[n][co][(y+oy)*out_pitch + (x+ox)]  for y < Hp-2, x < Wp-2
//   out[q] = sum_{ci,dy,dx} w[co][ci][dy][dx] * in[ci][q + dy*Wp + dx],  q = y*Wp + x
// so a tap is a pure linear offset and a block owns a contiguous run of q.
enum ConvEpilogue : int {
    EPI_RAW = 0,        // y (ConvGRU gates)                                   + GN quad stats
    EPI_SSE = 1,        // y * sigmoid(sum_co k1[co] y[co]) (ConvGRU candidate) + GN quad stats
    EPI_SWISH = 2,      // partial-conv ratio, swish                            + GN quad stats
    EPI_BIAS = 3,       // y + b                      (DSen2)
    EPI_BIAS_RELU = 4,  // relu(y + b)                (DSen2)
    EPI_BIAS_RES = 5,   // res + 0.1 * (y + b)        (DSen2 residual blocks)
    EPI_BIAS_TANH_ADD = 6  // res + tanh(y + b)       (DSen2 head; res = bilinear bands)
};

struct ConvSeg {
    const float* base;   // segment tensor
    long stride_n;       // floats between consecutive n (within a weight set)
    long set_off[2];     // extra offset for weight-set 0 / 1 (ConvGRU fw / bw frame choice)
    int C;               // channels in this segment
};

struct ConvArgs {
    ConvSeg seg[2];
    int Cin;             // seg[0].C + seg[1].C
    int Hp, Wp;          // padded input plane
    int Cout;            // real output channels
    const float* w;      // packed: [set][cout_block][chunk][tap][CK][BN]
    long w_set_stride;   // floats between weight sets
    int n_per_set;       // n / n_per_set selects the weight set (and seg.set_off)
    float* out;
    long out_stride_n, out_plane;
    int out_pitch, oy, ox;
    float* stats;        // [n][Cout/4][nblk_q] float2 (sum, sumsq) or nullptr
    const float* aux;    // EPI_SSE: k1[Cout]; EPI_BIAS*: bias[Cout]
    long aux_set_stride; // floats between the aux vectors of weight sets
    const float* res;    // EPI_BIAS_RES / _TANH_ADD: residual, same indexing as out
    int same_pad;        // EPI_SWISH: 1 -> multiply by the partial-conv ratio (zero-padded SAME conv)
    int reflect_out;     // EPI_BIAS*: 1 -> also write the 1-px reflect rim of the (padded) output plane
    int cin_run;         // > 0: only the first cin_run input channels are non-zero (ConvGRU step 0: h = 0) -- a kernel MAY skip whole
                         // 8-channel chunks past them (the Winograd kernels run ceil(cin_run / 8) chunks); every channel of a chunk that
                         // does run is still READ, so the caller provides zeros there (model.hip clears the 7 state planes of chunk 2)
    unsigned long long* trace;   // probe aid (TTC_F32_TRACE): per-workgroup s_memtime stamps of the traced instantiation, else nullptr
    // EPI <= EPI_SWISH (the GroupNorm layers): the output ALWAYS keeps the input pitch -- out_pitch == Wp, oy == ox == 0,
    // out_plane == (Hp-2)*Wp: out[co][q] for the tile's own flat positions q, junk columns included -- so that a tile leaves
    // through an LDS transpose as fully coalesced 16-byte stores (conv_epilogue_flat); out_pitch / oy / ox are ignored there
};

struct PackedConv {      // device copy of one layer's packed weights
    float* d_w = nullptr;
    int Cin = 0, Cout = 0, CK = 0, BN = 0, nsets = 1;
    int nchunk = 0, ncb = 0;
    long set_stride = 0;
    int mode = 0;                 // ttc_config.precision: 0 = exact fp32 MFMA, 2 = fp16 / 3 = bf16 pairs on the 16-bit engine
    int form = 0;                 // ttc_config.fp32_conv_form: 0 = F(4x4) where it applies, 1 = F(2x2) at most, 2 = direct only
    // 16-bit engine (precision 2 = fp16, 3 = bf16): LDS-image weights hi | lo per chunk, see conv3x3_h16.hip
    uint4* d_wh = nullptr;
    int nchunk_h = 0;             // 8-channel blocks (padded per input segment)
    long set_stride_h = 0;        // 16-byte units between weight sets
    int terms = 3;                // MFMA products per K block: 1 (hi*hi) or 3 (lo*hi + hi*lo + hi*hi)
    // fp32 engine, Winograd F(2x2, 3x3) form of the GroupNorm layers (conv3x3_wino.hip): U = G g G^T images
    float* d_wu = nullptr;
    int nchunk_w = 0;             // 8-channel chunks
    long set_stride_w = 0;        // floats between weight sets
    // ... and the Winograd F(4x4, 3x3) form of the layers with Cout % 64 == 0 (conv3x3_wino4.hip)
    float* d_wu4 = nullptr;
    int nchunk_w4 = 0;
    long set_stride_w4 = 0;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember the size configured on each device
// (one process normally drives one GPU, but two contexts on different devices must both work)
struct LdsConfig {
    size_t per_dev[64] = {0};
    template <class F>
    hipError_t ensure(F* fn, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        size_t& cur = per_dev[dev & 63];
        if (bytes > cur) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) return e;
            cur = bytes;
        }
        return hipSuccess;
    }
};

// ----------------------------------------------------------------------------- 16-bit conv engine (conv3x3_h16.hip)
// Activations that feed a convolution live CHANNEL-BLOCKED in HBM: [n][C8][Hp*Wp][8] 16-bit elements (fp16 or bf16),
// i.e. one 16-byte K vector per (channel block, position), as a `hi` tensor and an optional `lo` tensor with
// x = hi + lo (fp16 pair: 22 mantissa bits; bf16 pair: 16).  A conv stage then is a pure global -> LDS DMA.
// Pad channels (C not a multiple of 8) hold zeros.  All strides / offsets below are in 16-byte units.
struct H16Seg {
    const uint4* hi;     // [n][C8][PP]
    const uint4* lo;     // same layout, or nullptr (1-term layers read hi only)
    long stride_n;       // units between consecutive n (within a weight set)
    long set_off[2];     // extra offset for weight-set 0 / 1
    int C8;              // channel blocks in this segment
};
struct H16Args {
    H16Seg seg[2];
    int nchunk;          // chunks to run: seg[0].C8 + seg[1].C8, or seg[0].C8 alone when the second segment is identically zero (ConvGRU step 0)
    int nchunk_pack;     // chunks per (set, cout block) of the packed weight images (their stride); 0 = nchunk
    const uint4* w;      // packed LDS images: [set][cout_block][chunk][hi|lo][plane_units]
    long w_set_stride;
    // channel-blocked output (OUT_B16 kernels): [n][Cout8][o_plane] hi (+ lo), pixel (y + c.oy) * c.out_pitch + x + c.ox for the
    // DSen2 layers (16-bit pair of the next layer's input); GroupNorm layers (EPI <= EPI_SWISH): o_hi / o_lo = the planes of top /
    // bottom 16-bit halves of the raw fp32 output at the tile's flat positions q, o_plane = (Hp - 2) * Wp
    uint4* o_hi; uint4* o_lo; long o_stride_n; long o_plane;
    const uint4* r_hi; const uint4* r_lo;   // EPI_BIAS_RES residual in the same blocked layout / indexing as the output
    ConvArgs c;          // geometry (Hp, Wp, Cout, n_per_set) and the fp32 epilogue operands (out, stats, aux, res, ...)
};
enum H16Out : int { OUT_F32 = 0, OUT_B16 = 1 };

int conv_pick_ck(int Cin);
int conv_pick_bn(int Cout);
// packs HWIO host kernels (one per weight set) into the layout above; returns floats per set
long conv_pack(const float* const* hwio, int nsets, int Cin, int Cout, int CK, int BN, std::vector<float>& out);
int conv_q_blocks(int Hp, int Wp);   // 512-position tiles of a plane
int conv_stat_slots(int Hp, int Wp); // GroupNorm partial sums per (window, channel quad): one per tile and wave
hipError_t conv_launch(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s);
// flops of the MATRIX INSTRUCTIONS the launch conv_launch() makes for these arguments issues: workgroup tiles x k-steps x flops per MFMA, with
// the channel / region / cout padding the kernel really multiplies (Winograd forms: 2.25 or 4 multiply-accumulates per output and tap set
// instead of 9).  bench.py's roofline prices kernels with it (ttc_debug_kernel_flops); the algorithmic 2*9*Cin*Cout per pixel stays beside it.
double conv_issued_flops(const ConvArgs& a, const PackedConv& pw, int epi, int n);
double conv_issued_flops_h16(const H16Args& a, const PackedConv& pw, int n);
// Winograd F(2x2, 3x3) kernels of the fp32 engine (conv3x3_wino.hip): EPI_RAW / EPI_SSE / EPI_SWISH layers with Cout % 32 == 0
long conv_pack_wino(const float* const* hwio, int nsets, int Cin, int Cout, std::vector<float>& out, int* nchunk);
hipError_t conv_launch_wino(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s);
int conv_wino_stat_slots(int Hp, int Wp, int Cout);
// Winograd F(4x4, 3x3) kernels (conv3x3_wino4.hip): EPI_RAW / EPI_SWISH layers with Cout % 64 == 0
long conv_pack_wino4(const float* const* hwio, int nsets, int Cin, int Cout, std::vector<float>& out, int* nchunk);
hipError_t conv_launch_wino4(const ConvArgs& a, const PackedConv& pw, int epi, int n, hipStream_t s);
int conv_wino4_stat_slots(int Hp, int Wp);
bool conv_wino4_ok(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set);   // launch limits of the F(4x4) kernels
// which fp32 kernel conv_launch runs for a layer: ONE decision shared by the launch, the GroupNorm slot count and the ConvGRU step-0 shortcut
enum ConvKernel : int { CONV_DIRECT = 0, CONV_WINO2 = 1, CONV_WINO4 = 2 };
ConvKernel conv_kernel_for(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set);
int conv_stat_slots_for(const PackedConv& pw, int epi, int Hp, int Wp, int Cin, int n, int n_per_set);   // GroupNorm partial-sum slots of that kernel
// C0: real channels of the first input segment (its blocks are padded to a multiple of 8 on their own); bf: bf16 elements
long conv_pack_h16(const float* const* hwio, int nsets, int Cin, int C0, int Cout, int BN, bool bf, std::vector<uint16_t>& out,
                   int* nchunk);
hipError_t conv_launch_h16(const H16Args& a, const PackedConv& pw, int mode, int epi, int out_kind, int n, hipStream_t s);   // mode: ttc_ctx::blk_mode()
uint16_t h16_from_float(float f, bool bf);
float h16_to_float(uint16_t h, bool bf);
struct ttc_ctx;
// packs + uploads the weights of one layer for the engine selected by ctx->cfg.precision (both images when 1)
// C0: real channels of the first of two concatenated input segments (-1: one segment); only the 16-bit engine needs it
ttc_status conv_upload(ttc_ctx* c, PackedConv& pc, const float* const* hwio, int nsets, int Cin, int Cout, int BN, int C0 = -1, int mode = -1);   // mode: engine the images are packed for (-1 = cfg.precision)

// ----------------------------------------------------------------------------- context
struct Timing {
    int level = 0;
    struct Rec { double ms = 0; int64_t n = 0; double flops = 0; int64_t nf = 0; };   // flops: matrix-instruction flops ISSUED by the nf launches noted
    std::map<std::string, Rec> recs;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> pending;
};

struct ttc_ctx {
    ttc_config cfg{};
    int device = 0;
    std::string err;
    size_t dev_bytes = 0;
    std::vector<void*> allocs;
    // TTC_GUARD: guard zones around every owned device buffer (ttc_debug_check_guards)
    struct Guarded { char* base; size_t user_bytes; std::string name; };
    std::map<void*, Guarded> guarded;            // user pointer -> allocation
    static size_t guard_bytes();                  // 0 = off
    void* guarded_malloc(size_t bytes, const std::string& name);
    void guarded_free(void* user);
    std::map<std::string, std::pair<float*, size_t>> named;   // debug-visible activations

    // model weights
    bool have_model = false, have_dsen2 = false;
    PackedConv w_gates, w_cand;                 // 2 sets (fw, bw)
    PackedConv w_block[8];                      // conv_median, conv_concat, conv1, conv2, up2, up2_out, up3, out
    float* d_small = nullptr;                   // small per-channel vectors (gamma/beta/sse/head ...)
    std::map<std::string, long> small_off;      // name -> offset (floats) into d_small
    PackedConv w_ds[6];
    float* d_ds_bias = nullptr;

    // workspace (model)
    float *frames = nullptr, *h[2] = {nullptr, nullptr}, *rh = nullptr, *yg = nullptr, *ug = nullptr,
          *yc = nullptr, *gru_out = nullptr;
    float *y_med = nullptr, *z_med = nullptr, *y_cat = nullptr, *p1 = nullptr, *y_c1 = nullptr, *p2 = nullptr,
          *y_c2 = nullptr, *u2in = nullptr, *y_u2 = nullptr, *u2a = nullptr, *y_u2o = nullptr,
          *u3in = nullptr, *y_u3 = nullptr, *oa = nullptr, *y_out = nullptr;
    // 16-bit engine (cfg.precision >= 2): channel-blocked hi / lo activations [n][C8][plane] (16-byte units), see conv3x3_h16.hip
    struct B16 { uint4* hi = nullptr; uint4* lo = nullptr; };
    B16 frames16, h16[2], rh16, gru16, z_med16, p1_16, p2_16, u2in16, u2a16, u3in16, oa16;
    float *stats = nullptr, *gn = nullptr;      // GN partial sums / (mean, rstd)
    int cd_wins_T = 0;                          // cloud detection: the date-window table on the device was built for this many dates (0 = none)
    size_t stats_floats = 0;
    int clouds_debug_stage = 0;   // ttc_debug_clouds_stage: return the flags after that stage of the cloud detector (test aid)
    bool keep_debug = false;      // ttc_debug_keep: also materialise intermediates that the fused kernels never write (test aid)
    // workspace (tile-level), grown on demand
    std::map<std::string, std::pair<void*, size_t>> scratch;
    std::map<std::string, std::pair<void*, size_t>> pinned;    // page-locked host staging, same keyed-growth scheme
    int* spec_status = nullptr;   // single-call tile path: device int32[4] the speculative stages report into (see ttc_predict_tile)
    unsigned wmat_slot = 0;       // ring index of the host-built temporal operator (tile.hip)
    std::vector<hipEvent_t> wmat_events;   // per ring slot: the H2D copy that last read it
    bool want_planar_frames = false;   // the caller asked for the model feed (ttc_predict_tile d_model_in): keep the fp32 planar frames
    // which form of the model frames the LAST writer left valid: every writer states it and hands it to model_forward_frames
    // explicitly (a writer cannot "forget" a flag: the parameter is required); ttc_debug_fetch("frames") refuses a stale planar buffer
    bool frames_planar_valid = false;
    int forward_n = 0;            // windows of the last forward: the raw split planes of its last block were laid out with this n (model_taps)
    bool minv_ready = false;      // the constant Whittaker matrix has been uploaded (tile.hip)

    Timing timing;

    ttc_status fail(ttc_status s, const std::string& m) { err = m; return s; }
    float* alloc_f(size_t n, const char* name = nullptr);
    bool alloc_b16(B16& b, size_t units);       // hi + lo tensors of `units` 16-byte K vectors each
    bool half() const { return cfg.precision >= 2; }      // 16-bit conv engine selected (channel-blocked hi / lo pairs)
    int blk_mode() const { return cfg.precision == 3 ? 1 : 0; }   // Elem<> mode: fp16 / bf16
    // DSen2's convs may run on the 16-bit engine (hi + lo pairs, three products) inside an fp32 context: ttc_config.dsen2_precision
    bool ds_half() const { return half() || cfg.dsen2_precision >= 2; }
    int ds_blk_mode() const { return half() ? blk_mode() : (cfg.dsen2_precision == 3 ? 1 : 0); }
    void* scratch_buf(const std::string& key, size_t bytes);
    void* pinned_buf(const std::string& key, size_t bytes);
};

// adjust_shape (src/download_and_predict_job.py:260-310) as an index map: element (i, j) of the [width, height] result reads element
// (clamp(i + o1, 0, n1 - 1), clamp(j + o2, 0, n2 - 1)) of the [n1, n2] source -- an edge pad is a negative offset, a crop a positive one.
struct AdjustMap { int n1, n2, o1, o2; };
// false where the reference's rule does not produce `want` (odd differences of 3 or more: its own result keeps the wrong length and the next
// statement of process_tile raises on the broadcast)
inline bool adjust_axis(int n, int want, int* off) {
    *off = 0;
    if (n < 1 || want < 1) return false;
    if (n == want) return true;
    const int d = n < want ? want - n : n - want, amt = d / 2;
    if (amt != 0 && (d & 1)) return false;
    *off = (n < want ? -1 : 1) * (amt == 0 ? 1 : amt);       // pad: (1, 0) or (amt, amt), :271-283; crop: [1:] or [amt:-amt], :285-307
    return true;
}
inline bool adjust_map(int n1, int n2, int width, int height, AdjustMap* m) {
    m->n1 = n1; m->n2 = n2;
    return adjust_axis(n1, width, &m->o1) && adjust_axis(n2, height, &m->o2);
}

// model.hip
ttc_status model_alloc(ttc_ctx* c);
ttc_status model_load(ttc_ctx* c, const ttc_tensor* t, int n);
enum FramesForm { FRAMES_PLANAR = 0,   // c->frames holds the fp32 planar padded frames (the 16-bit engine converts them first)
                  FRAMES_B16 = 1 };    // c->frames16 holds the channel-blocked hi / lo pairs (tile.hip k_assemble<BF>); c->frames is stale
ttc_status model_forward_frames(ttc_ctx* c, int n, float* d_out, hipStream_t s, FramesForm form);
ttc_status model_frames_from_nhwc(ttc_ctx* c, const float* d_in, int n, hipStream_t s);
ttc_status model_taps(ttc_ctx* c, int n, float* d_early, float* d_late, hipStream_t s);
// reseg.hip
ttc_status reseg_border_subtiles(ttc_ctx* c, const float* d_s2, const float* d_s1, const float* d_dem, int X,
                                 const int32_t* h_rows, int n, const float* h_min, const float* h_max, int hist_align,
                                 int n_dates_ok, float* d_preds, float* h_stats, int32_t* h_applied, hipStream_t s);
ttc_status reseg_mosaic(ttc_ctx* c, const float* d_preds, const ttc_reseg_window* h_wins, int n, const float* d_weights,
                        const double* d_ramps, int X, int Y, float* d_out, float* d_sums, hipStream_t s);
ttc_status reseg_seam_adjust(ttc_ctx* c, float* d_preds, int n, int oh, int ow, float* h_stats, hipStream_t s);
// dsen2.hip
ttc_status dsen2_load(ttc_ctx* c, const ttc_tensor* t, int n);

// timing helper: wraps a launch in HIP events on stream s when enabled
struct KTimer {
    ttc_ctx* c; const char* name; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    // conv launches: the flops of the matrix instructions the launch issues (conv_issued_flops), summed per family next to the event times
    void flops(double f) { if (a) { auto& r = c->timing.recs[name]; r.flops += f; r.nf += 1; } }
    KTimer(ttc_ctx* c_, const char* n_, hipStream_t s_) : c(c_), name(n_), s(s_) {
        // level 1: every kernel family; level 2: only the conv engine launches (names "conv_*", "dsen2_conv")
        if (c->timing.level == 1 || (c->timing.level == 2 && strstr(name, "conv") != nullptr)) {
            hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s);
        }
    }
    ~KTimer() {
        if (a) { hipEventRecord(b, s); c->timing.pending.push_back({name, {a, b}}); }
    }
};
