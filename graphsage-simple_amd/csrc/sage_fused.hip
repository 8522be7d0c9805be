// One-launch GraphSAGE layer: gather-mean tile in LDS -> fp32 MFMA -> act.
// (first milestone: dispatch only; kernels follow)
#include "sage_internal.h"

bool sage_layer_fused_supported(int32_t dim, int32_t out_dim, int32_t concat) {
    (void)dim; (void)out_dim; (void)concat;
    return false;
}

int sage_launch_layer_fused(const float*, int64_t, int64_t, int32_t, const int32_t*, const int32_t*, int32_t, int32_t,
                            const int32_t*, const int32_t*, const int32_t*, const int32_t*, int32_t, const int32_t*,
                            const float*, int64_t, int32_t, int32_t, float*, int64_t, hipStream_t) {
    sage_set_error("layer_forward: no fused kernel for this shape");
    return SAGE_EUNSUPPORTED;
}

extern "C" int sage_layer_forward_supported(int32_t dim, int32_t out_dim, int32_t concat) {
    return sage_layer_fused_supported(dim, out_dim, concat) ? 1 : 0;
}

extern "C" int sage_layer_forward(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                                  const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                                  const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                                  const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                                  sage_stream_t stream) {
    SAGE_REQUIRE(table && nbr && cnt && weight && out, "layer_forward: NULL array");
    SAGE_REQUIRE(n >= 0 && k >= 1 && dim >= 1 && out_dim >= 1, "layer_forward: n=%d k=%d dim=%d out_dim=%d", n, k, dim, out_dim);
    SAGE_REQUIRE(ld >= dim && ldo >= out_dim && ldw >= (concat ? 2 : 1) * (int64_t)dim, "layer_forward: leading dimensions");
    SAGE_REQUIRE(table_rows >= 1 && table_rows < (1ll << 31), "layer_forward: table_rows = %lld", (long long)table_rows);
    SAGE_REQUIRE(act >= 0 && act <= SAGE_ACT_NONE, "layer_forward: act = %d", act);
    return sage_launch_layer_fused(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, concat,
                                   self_index, weight, ldw, out_dim, act, out, ldo, (hipStream_t)stream);
}
