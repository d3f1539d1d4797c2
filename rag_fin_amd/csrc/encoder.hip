// Embedder entry points (filled in by the encoder kernels; this revision only
// reserves the ABI so that the library exports every symbol of ragfin.h).
#include "rf_internal.h"

struct rf_encoder { int unused; };

extern "C" int rf_encoder_create(rf_encoder_t** out, const rf_encoder_config*, const rf_encoder_weights*, int) {
  if (out) *out = nullptr;
  rf_set_error("rf_encoder_create: encoder kernels are not part of this build yet");
  return RF_ERR_UNSUPPORTED;
}
extern "C" int rf_encoder_destroy(rf_encoder_t*) { return RF_OK; }
extern "C" size_t rf_encode_workspace_bytes(const rf_encoder_t*, int, int) { return 0; }
extern "C" int rf_encode(const rf_encoder_t*, const int32_t*, const int32_t*, int, int, void*, float*,
                         void*, size_t, void*) {
  rf_set_error("rf_encode: encoder kernels are not part of this build yet");
  return RF_ERR_UNSUPPORTED;
}
