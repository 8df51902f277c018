// placeholder until the conv engine lands
#include "mm_common.h"
extern "C" {
int64_t mm_resnet50_blob_floats(void) { return 0; }
int mm_resnet50_create(mm_resnet50_t** out, const float*, int64_t, int, int, float) { if (out) *out = nullptr; return MM_ERR_UNSUPPORTED; }
int mm_resnet50_destroy(mm_resnet50_t*) { return MM_OK; }
int64_t mm_resnet50_workspace_bytes(mm_resnet50_t*, int64_t) { return MM_ERR_UNSUPPORTED; }
int mm_resnet50_forward(mm_resnet50_t*, const float*, int, int64_t, float*, void*, int64_t, void*) { return MM_ERR_UNSUPPORTED; }
int64_t mm_head_blob_floats(void) { return 0; }
int mm_head_create(mm_head_t** out, const float*, int64_t) { if (out) *out = nullptr; return MM_ERR_UNSUPPORTED; }
int mm_head_destroy(mm_head_t*) { return MM_OK; }
int64_t mm_head_workspace_bytes(mm_head_t*, int64_t, int64_t) { return MM_ERR_UNSUPPORTED; }
int mm_head_forward(mm_head_t*, const float*, const float*, int, const float*, int64_t, int64_t, float*, void*, int64_t, void*) { return MM_ERR_UNSUPPORTED; }
}
