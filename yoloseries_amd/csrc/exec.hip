// Program executor: the engine (yoloseries_amd/engine.py) describes a forward / backward pass once as an array of commands —
// entry point of this library + its arguments widened to 8-byte slots, or an event record / stream wait — and replays it with
// ONE call.  Per launch the host then pays the HIP launch only (~3 us) instead of a Python iteration + a ctypes call (~7 us):
// at small batch the step is bound by exactly that.  The commands call the same extern "C" functions a host
// would call one by one; nothing is captured or reordered (unlike a hipGraph, whose kernel nodes were measured slower).
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

template <typename T> inline T slot_as(const uint64_t s)
{
    if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(static_cast<uintptr_t>(s));
    else if constexpr (std::is_floating_point_v<T>) { double v; memcpy(&v, &s, 8); return static_cast<T>(v); }
    else return static_cast<T>(static_cast<int64_t>(s));
}
template <typename... P, size_t... I>
inline int call_slots(int (*fn)(P...), const uint64_t* s, std::index_sequence<I...>) { return fn(slot_as<P>(s[I])...); }
template <typename... P>
inline int thunk_impl(int (*fn)(P...), const uint64_t* s) { return call_slots(fn, s, std::index_sequence_for<P...>{}); }
template <typename... P> constexpr int arity(int (*)(P...)) { return (int)sizeof...(P); }

struct OpEntry { const char* name; int (*thunk)(const uint64_t*); int nargs; };
#define YH_OP(fn) { #fn, [](const uint64_t* s) -> int { return thunk_impl(&fn, s); }, arity(&fn) }
// every entry point a Program command list may contain; the stream is each function's LAST argument
const OpEntry kOps[] = {
    YH_OP(yh_conv_igemm), YH_OP(yh_conv_wgrad), YH_OP(yh_bn_finalize), YH_OP(yh_bn_fold_batch), YH_OP(yh_bn_silu_apply),
    YH_OP(yh_bn_silu_bwd_reduce), YH_OP(yh_bn_bwd_finalize), YH_OP(yh_bn_silu_bwd_apply), YH_OP(yh_colsum),
    YH_OP(yh_maxpool5_fwd), YH_OP(yh_maxpool5_bwd), YH_OP(yh_upsample2_bwd), YH_OP(yh_fill_u32),
    YH_OP(yh_bn_silu_apply_parts), YH_OP(yh_bn_silu_bwd_apply_parts), YH_OP(yh_bn_finalize_parts), YH_OP(yh_bn_bwd_finalize_parts),
    YH_OP(yh_bn_frozen), YH_OP(yh_sppf_pool3_fwd), YH_OP(yh_sppf_pool3_bwd),
};
constexpr int kNumOps = (int)(sizeof(kOps) / sizeof(kOps[0]));

}  // namespace

/* index of an entry point in the executor's table (-1: not executable through yh_exec) and its argument count incl. the stream */
extern "C" int yh_exec_op(const char* name, int* nargs)
{
    for (int i = 0; i < kNumOps; ++i)
        if (name && strcmp(name, kOps[i].name) == 0) { if (nargs) *nargs = kOps[i].nargs; return i; }
    return -1;
}

extern "C" int yh_exec(const yh_cmd* cmds, int n, const yh_stream* streams, int nstreams, int* failed)
{
    YH_CHECK_ARG(cmds && n >= 0 && streams && nstreams >= 1, "yh_exec: bad args");
    // timing-only diagnostics (results wrong: the streams race): YH_EXEC_ABL=1 drops the event records / stream waits of a program
    static const int abl = [] { const char* e = getenv("YH_EXEC_ABL"); return e ? atoi(e) : 0; }();
    for (int i = 0; i < n; ++i) {
        const yh_cmd& c = cmds[i];
        int rc = YH_OK;
        if (c.stream < 0 || c.stream >= nstreams) rc = YH_EINVAL;
        else if (abl && (c.op == YH_CMD_EVENT_RECORD || c.op == YH_CMD_STREAM_WAIT)) {
        } else if (c.op == YH_CMD_EVENT_RECORD) {
            if (hipEventRecord((hipEvent_t)(uintptr_t)c.slots[0], (hipStream_t)streams[c.stream]) != hipSuccess) rc = YH_ELAUNCH;
        } else if (c.op == YH_CMD_STREAM_WAIT) {
            if (hipStreamWaitEvent((hipStream_t)streams[c.stream], (hipEvent_t)(uintptr_t)c.slots[0], 0) != hipSuccess) rc = YH_ELAUNCH;
        } else if (c.op >= 0 && c.op < kNumOps && c.nslots == kOps[c.op].nargs && c.nslots <= YH_CMD_SLOTS) {
            uint64_t s[YH_CMD_SLOTS];
            memcpy(s, c.slots, sizeof(uint64_t) * c.nslots);
            s[c.nslots - 1] = (uint64_t)(uintptr_t)streams[c.stream];
            rc = kOps[c.op].thunk(s);
        } else rc = YH_EINVAL;
        if (rc != YH_OK) {
            if (failed) *failed = i;
            if (rc == YH_EINVAL && (c.op < YH_CMD_STREAM_WAIT || c.op >= kNumOps)) yh_set_error("yh_exec: command %d is malformed (op %d, %d slots)", i, c.op, c.nslots);
            return rc;
        }
    }
    return YH_OK;
}
