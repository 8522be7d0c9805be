// Backward kernels of linear_act / gather_mean (training harness).
#include "sage_internal.h"

extern "C" int sage_linear_act_backward(const float*, int64_t, const int32_t*, const float*, int64_t, int32_t, const float*,
                                        int64_t, int32_t, int32_t, const float*, int64_t, const float*, int64_t, int32_t,
                                        const int32_t*, float*, int64_t, float*, int64_t, sage_stream_t) {
    sage_set_error("linear_act_backward: not built yet");
    return SAGE_EUNSUPPORTED;
}

extern "C" int sage_gather_mean_backward(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int32_t, int32_t,
                                         const int32_t*, const int32_t*, const int32_t*, float*, int64_t, int64_t,
                                         sage_stream_t) {
    sage_set_error("gather_mean_backward: not built yet");
    return SAGE_EUNSUPPORTED;
}
