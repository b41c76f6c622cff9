// csrc/sift.hip -- placeholder, replaced by the SIFT kernels
#include "common.h"
int mi_sift_extract_dev(mi355_ctx* ctx, int, const uint8_t*, int, int, int, int*) { ctx->set_error("sift: not built yet"); return MI355_ERR_FAILED; }
void mi_sift_release(mi355_ctx*) {}
