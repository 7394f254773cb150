// placeholder until the MFMA pairwise tile lands (fails loudly; there is no fallback)
#include "common.h"
extern "C" size_t pope_pairwise_scratch_bytes(int64_t, int32_t, int32_t) { return 0; }
extern "C" int pope_pairwise_minmax(const float *, int64_t, int32_t, const float *, int32_t, int32_t, float *, int64_t,
                                    int32_t, void *, size_t, void *) {
    pope::set_error("pope_pairwise_minmax: not built yet");
    return POPE_ERR_INVALID;
}
