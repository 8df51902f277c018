// CU-partitioned HIP streams.  The hot path alternates matrix-core-bound GEMMs with HBM-bound, zero-FLOP kernels (Winograd input
// transforms, max-pool, preprocessing); on ordinary streams the hardware dispatcher fills all 256 CUs with whichever grid arrives
// first, so two lanes only overlap at their tails.  A stream created with a CU mask confines its kernels to a subset of the CUs,
// which lets the pipeline pin the movement kernels of one lane to a small partition while another lane's GEMMs own the rest.
// No reference counterpart (the reference runs on torch's default stream only, api/steerable/utils.py:34-50).
#include "mm_common.h"

extern "C" {

int mm_stream_create_cu_mask(void** stream, const uint32_t* mask, int words) {
    if (!stream || !mask || words <= 0 || words > 64) return MM_ERR_INVALID_ARG;
    *stream = nullptr;
    bool any = false;
    for (int i = 0; i < words; ++i) any = any || mask[i] != 0;
    if (!any) return MM_ERR_INVALID_ARG;       // a stream no CU may serve would hang its first kernel
    hipStream_t s = nullptr;
    MM_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    *stream = (void*)s;
    return MM_OK;
}

int mm_stream_get_cu_mask(void* stream, uint32_t* mask, int words) {
    if (!mask || words <= 0 || words > 64) return MM_ERR_INVALID_ARG;
    MM_HIP(hipExtStreamGetCUMask((hipStream_t)stream, (uint32_t)words, mask));
    return MM_OK;
}

int mm_stream_destroy(void* stream) {
    if (!stream) return MM_OK;
    MM_HIP(hipStreamDestroy((hipStream_t)stream));
    return MM_OK;
}

}  // extern "C"
