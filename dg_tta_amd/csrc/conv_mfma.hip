// MFMA implicit-GEMM conv kernels (placeholder until the tiled kernels land: reports "unsupported" so that the
// dispatcher in unet_ref.hip uses the general VALU kernels).
#include "common.h"

int conv3_fwd_mfma(const void *, int, const void *, const float *, void *, int, void *, int, int, int, int, int, int, int,
                   int, int, int, hipStream_t) {
  return DGTTA_ERR_UNSUPPORTED;
}
int conv3_wgrad_mfma(const void *, int, const void *, int, float *, float *, void *, size_t, int, int, int, int, int, int,
                     int, int, int, hipStream_t) {
  return DGTTA_ERR_UNSUPPORTED;
}
