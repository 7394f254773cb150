// placeholder until the SAGEConv kernels land (fails loudly; there is no fallback)
#include "common.h"
extern "C" size_t sage_conv_scratch_bytes(int64_t, int64_t, int64_t, int32_t, int32_t) { return 0; }
extern "C" int sage_conv_forward(const int32_t *, const int32_t *, int64_t, int64_t, int64_t, const float *, int32_t,
                                 const float *, const float *, const float *, int32_t, float *, float *, void *) {
    pope::set_error("sage_conv_forward: not built yet");
    return POPE_ERR_INVALID;
}
extern "C" int sage_conv_backward(const int32_t *, const int32_t *, int64_t, int64_t, int64_t, const float *,
                                  const float *, int32_t, const float *, const float *, int32_t, const float *, float *,
                                  float *, float *, float *, void *, size_t, void *) {
    pope::set_error("sage_conv_backward: not built yet");
    return POPE_ERR_INVALID;
}
