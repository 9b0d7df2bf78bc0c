// TEMPORARY: entry points not built yet return TTC_ERR_STATE.
#include "ttc_internal.h"
#define NB(c) (c)->fail(TTC_ERR_STATE, std::string(__func__) + ": not built yet")
ttc_status tile_process_subtiles(ttc_ctx* c, const float*, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, int, int, float*, float*, hipStream_t) { return NB(c); }
ttc_status tile_missing_counts(ttc_ctx* c, const float*, int, int, int, int32_t*, hipStream_t) { return NB(c); }
ttc_status tile_fix_missing(ttc_ctx* c, float*, int, int, int, int, int, hipStream_t) { return NB(c); }
ttc_status mosaic_run(ttc_ctx* c, const float*, int, const int32_t*, int, int, int, uint8_t*, float*, hipStream_t) { return NB(c); }
ttc_status dsen2_forward(ttc_ctx* c, const float*, const float*, int, int, int, float*, hipStream_t) { return NB(c); }
ttc_status dsen2_tile(ttc_ctx* c, float*, int, int, int, int, hipStream_t) { return NB(c); }
ttc_status upsample_20m(ttc_ctx* c, const float*, const float*, int, int, int, float*, hipStream_t) { return NB(c); }
ttc_status dsen2_load(ttc_ctx* c, const ttc_tensor*, int) { return NB(c); }
