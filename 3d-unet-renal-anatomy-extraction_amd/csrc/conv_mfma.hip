// bf16 MFMA implicit-GEMM convolution kernels (placeholder: not yet enabled).
#include "common.h"
#include "conv.h"

bool mfma_conv_eligible(int, int, int, int, int) { return false; }
size_t mfma_packed_bytes(int, int, int) { return 0; }
int pack_mfma_launch(const float*, void*, int, int, int, int64_t, int64_t, int, hipStream_t) {
    return ru3d_fail(-1, "mfma path not built");
}
int conv_mfma_launch(const void*, const void*, const float*, const void*, void*, const ConvGeom&, hipStream_t) {
    return ru3d_fail(-1, "mfma path not built");
}
bool mfma_wgrad_eligible(const WgradGeom&, int) { return false; }
size_t wgrad_mfma_ws_bytes(const WgradGeom&) { return 0; }
int wgrad_mfma_launch(const void*, const void*, float*, void*, size_t, WgradGeom, hipStream_t) {
    return ru3d_fail(-1, "mfma path not built");
}
