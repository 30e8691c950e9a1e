// TEMPORARY: entry points not yet implemented (replaced by loss_v5.hip / postproc.hip).
#include "common.h"
#define STUB(sig) extern "C" int sig { yh_set_error("not implemented yet"); return YH_EUNSUPPORTED; }
extern "C" size_t yh_v5loss_ws_bytes(const yh_v5loss_desc*) { return 0; }
extern "C" size_t yh_v5loss_saved_bytes(const yh_v5loss_desc*) { return 0; }
STUB(yh_v5_assign(const yh_v5loss_desc*, const float*, int32_t*, float*, int32_t*, void*, yh_stream))
STUB(yh_v5_loss_fwd(const yh_v5loss_desc*, const void* const*, const float*, float*, float*, void*, void*, yh_stream))
STUB(yh_v5_loss_bwd(const yh_v5loss_desc*, const void* const*, const float*, const void*, void* const*, void*, yh_stream))
STUB(yh_iou_matrix(const float*, int, const float*, int, float, float*, yh_stream))
STUB(yh_iou_pairwise(int, const float*, const float*, int, float*, float*, yh_stream))
STUB(yh_decode_full(const yh_decode_desc*, const void* const*, float*, yh_stream))
STUB(yh_decode_filter(const yh_decode_desc*, const void* const*, float, float, float*, int32_t*, int, yh_stream))
extern "C" size_t yh_nms_ws_bytes(int, int) { return 0; }
STUB(yh_nms_batched(const float*, const int32_t*, int, int, float, int, int, int, int, float*, int32_t*, int32_t*, void*, yh_stream))
