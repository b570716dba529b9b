// Launch lists: a recorded sequence of entry-point calls, event records and stream waits, re-issued by one call
// (include/pfa_hip.h, "launch lists").  Host code only.
#include <string.h>

#include <tuple>
#include <type_traits>

#include "pfa_common.h"

namespace {

// one argument out of a blob laid out like a C struct of the argument types
template <typename T> T take(const char* blob, size_t& off) {
    off = (off + alignof(T) - 1) & ~(alignof(T) - 1);
    T v;
    memcpy(&v, blob + off, sizeof(T));
    off += sizeof(T);
    return v;
}

template <typename... A> int call_packed(int (*f)(A...), const char* blob) {
    size_t off = 0;
    std::tuple<std::decay_t<A>...> args{take<std::decay_t<A>>(blob, off)...};       // braced list: left to right
    return std::apply(f, args);
}

template <auto F> int thunk(const void* blob) { return call_packed(F, (const char*)blob); }

struct Entry {
    const char* name;
    int (*fn)(const void*);
};
#define PFA_ENTRY(f) {#f, &thunk<&f>}
const Entry g_table[] = {
    PFA_ENTRY(pfa_gather_gemm), PFA_ENTRY(pfa_splitk_finish), PFA_ENTRY(pfa_wgrad), PFA_ENTRY(pfa_tconv_scatter),
    PFA_ENTRY(pfa_logit_fwd), PFA_ENTRY(pfa_logit_dgrad), PFA_ENTRY(pfa_logit_wgrad),
    PFA_ENTRY(pfa_bn_finalize), PFA_ENTRY(pfa_bn_eval_coeffs), PFA_ENTRY(pfa_bn_act), PFA_ENTRY(pfa_bn_bwd_reduce),
    PFA_ENTRY(pfa_bn_bwd_finalize), PFA_ENTRY(pfa_bn_bwd_apply), PFA_ENTRY(pfa_bce_logits), PFA_ENTRY(pfa_l1_tanh_bwd),
    PFA_ENTRY(pfa_colsum), PFA_ENTRY(pfa_adam), PFA_ENTRY(pfa_adam_tick), PFA_ENTRY(pfa_weight_shadow),
    PFA_ENTRY(pfa_weight_shadow_batch), PFA_ENTRY(pfa_pack_nhwc), PFA_ENTRY(pfa_unpack_nchw), PFA_ENTRY(pfa_unpack_u8),
    PFA_ENTRY(pfa_fill_zero), PFA_ENTRY(pfa_axpy), PFA_ENTRY(pfa_finish_bn_fwd), PFA_ENTRY(pfa_finish_bn_bwd),
    PFA_ENTRY(pfa_bce_logits_parts), PFA_ENTRY(pfa_l1_tanh_bwd_parts), PFA_ENTRY(pfa_loss_finish),
    PFA_ENTRY(pfa_bn_act_stats), PFA_ENTRY(pfa_bn_bwd_apply_stats), PFA_ENTRY(pfa_set_knob), PFA_ENTRY(pfa_logit_fwd_bce), PFA_ENTRY(pfa_adam_shadow_batch),
    PFA_ENTRY(pfa_allreduce), PFA_ENTRY(pfa_finish_coop_fwd), PFA_ENTRY(pfa_finish_coop_bwd),
};
constexpr int g_entries = sizeof(g_table) / sizeof(g_table[0]);

}  // namespace

extern "C" int pfa_replay_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < g_entries; ++i)
        if (strcmp(g_table[i].name, name) == 0) return i;
    return -1;
}

extern "C" int pfa_replay(const pfa_replay_op* ops, int n, int* failed_at) {
    PFA_REQUIRE(ops || n == 0, "pfa_replay: null list");
    for (int i = 0; i < n; ++i) {
        const pfa_replay_op& op = ops[i];
        int rc = 0;
        switch (op.kind) {
            case PFA_OP_CALL:
                if (op.fn < 0 || op.fn >= g_entries || !op.a) {
                    if (failed_at) *failed_at = i;
                    PFA_FAIL(-1, "pfa_replay: op %d names no entry point", i);
                }
                rc = g_table[op.fn].fn(op.a);           // (its own message is in the error string)
                break;
            case PFA_OP_RECORD: {
                const hipError_t e = hipEventRecord((hipEvent_t)op.a, (hipStream_t)op.b);
                if (e != hipSuccess) { snprintf(g_pfa_err, sizeof(g_pfa_err), "pfa_replay: op %d: hipEventRecord: %s", i, hipGetErrorString(e)); rc = (int)e; }
                break;
            }
            case PFA_OP_WAIT: {
                const hipError_t e = hipStreamWaitEvent((hipStream_t)op.b, (hipEvent_t)op.a, 0);
                if (e != hipSuccess) { snprintf(g_pfa_err, sizeof(g_pfa_err), "pfa_replay: op %d: hipStreamWaitEvent: %s", i, hipGetErrorString(e)); rc = (int)e; }
                break;
            }
            default:
                if (failed_at) *failed_at = i;
                PFA_FAIL(-1, "pfa_replay: op %d has kind %d", i, op.kind);
        }
        if (rc != 0) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return 0;
}
