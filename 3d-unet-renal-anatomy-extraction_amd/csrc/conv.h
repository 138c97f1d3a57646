// Internal geometry structs and launcher prototypes shared by the conv translation units.
#pragma once
#include "common.h"

namespace RU3D_NS {

struct ConvGeom {
    int N;
    int Di, Hi, Wi, Cin, ldx;    // input tensor (of the data-movement form, not of the nn.Module)
    int Do, Ho, Wo, Cout, ldy;   // output tensor
    int CoutPad;                 // generic layout: padded cout of the packed weight
    int ldr;                     // residual pitch (when res != NULL)
    int k, stride, pad;
    int transposed;              // 0 = gather form, 1 = transposed (fractionally strided) form
    int zero_far;                // write exact zeros on the last plane of each output axis
    int flip;                    // read weight tap (taps-1-tap): stride-1 dgrad as a gather conv
    // split (planar concat) layouts, see ru3d_tensor: 0 = dense.  Only the kernels of a decoder ResBlock on the
    // full-resolution level take them (conv_slide64 input, conv_slide32 pair output)
    int x_cseg = 0, y_cseg = 0;
    int64_t x_segstride = 0, y_segstride = 0;
};

struct WgradGeom {
    int N;
    int Di, Hi, Wi, Cin, ldx;    // "x" operand (gathered with stride/pad)
    int Do, Ho, Wo, Cout, lddy;  // "dy" operand (dense positions)
    int k, taps, stride, pad;
    int64_t s_o, s_i;            // element strides of dw for (cout, cin); tap stride is 1
    int64_t chunk_len;           // positions per chunk (filled by the launcher)
    int x_cseg = 0;              // split (planar concat) x, see ru3d_tensor: 0 = dense
    int64_t x_segstride = 0;
};

// batched weight packing (conv_generic.hip): both packed layouts, up to RU3D_PACK_MAX weights per launch
struct PackOne {
    const float* src;
    void* dst;
    int cin, cout, taps, mfma, cout_pad;   // cin / cout: PACKED (kernel-side) channel counts; mfma: 0 generic, 1 MFMA, 2 bias
    int co_real, co_pad, ci_real, ci_pad;  // channel padding per segment (pad == 0: none), see pack_src_index
    int64_t s_o, s_i, total;
};
struct PackBatch {
    int count;
    PackOne item[RU3D_PACK_MAX];
};
int pack_batch_launch(const PackBatch& b, int dtype, hipStream_t st);

// Fused packing of the two MFMA roles of one weight (forward + input gradient) from ONE read of the fp32 source:
// the source is [a][b][tap] (Conv3d: a = cout, b = cin; ConvTranspose3d: a = cin, b = cout); dst_a is the packed form
// whose output-channel dimension is a (Conv forward / ConvTranspose input gradient), dst_b the one whose output-channel
// dimension is b.  Either may be NULL.  adim / bdim are the PACKED (padded) extents, multiples of 32.
struct PackPair {
    const float* src;
    void* dst_a;
    void* dst_b;
    int adim, bdim, taps;
    int a_real, a_pad, b_real, b_pad;   // channel padding per segment (pad == 0: none), see pack_src_index
    int64_t s_a;                        // source stride of a (= real bdim * taps); b has stride taps, tap stride 1
};
struct PackPairBatch {
    int count;
    PackPair item[RU3D_PACK_MAX];
};
int pack_pair_launch(const PackPairBatch& b, hipStream_t st);
int unpad_weight_launch(const float* src, float* dst, int cout, int cin, int taps, int co_real, int co_pad, int ci_real,
                        int ci_pad, int cin_p, hipStream_t st);

// conv_generic.hip
int generic_cot(int cout);
int generic_cout_pad(int cout);
int conv_generic_launch(const void* x, const void* w, const float* bias, const void* res, void* y,
                        const ConvGeom& g, int dtype, int y_dtype, hipStream_t st);
int pack_generic_launch(const float* src, void* dst, int cin, int cout, int taps, int64_t s_o, int64_t s_i,
                        int flip, int dtype, hipStream_t st);
size_t wgrad_generic_ws_bytes(const WgradGeom& g);
int wgrad_reduce_launch(const float* part, float* dw, int chunks, int taps, int cin, int cout, int64_t s_o, int64_t s_i,
                        hipStream_t st);
int wgrad_generic_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, WgradGeom g, int dtype,
                         hipStream_t st);

// conv_mfma.hip (bf16 MFMA implicit GEMM)
bool mfma_conv_eligible(int cin, int cout, int k, int dtype, int y_dtype);
bool mfma_conv_geometry_ok(const ConvGeom& g);
size_t mfma_packed_bytes(int cin, int cout, int taps);
int pack_mfma_launch(const float* src, void* dst, int cin, int cout, int taps, int64_t s_o, int64_t s_i, int flip,
                     hipStream_t st);
int conv_mfma_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                     hipStream_t st, float* stat_slab = nullptr, void* ws = nullptr, size_t ws_bytes = 0,
                     const void* bst_act = nullptr, int bst_ld = 0, float slope = 0.f, const void* x2 = nullptr,
                     int ldx2 = 0, const void* w2 = nullptr, int* defer_ks = nullptr);
// y = conv3_dgrad(x) + W2^T x2 in one launch: only the 32-channel sliding kernel has the 28th tap
bool mfma_conv_can_fuse_partner(const ConvGeom& g);
size_t conv_mfma_ws_bytes(const ConvGeom& g);
bool mfma_conv_can_fuse_stats(const ConvGeom& g);
size_t mfma_conv_stats_slab_bytes(const ConvGeom& g);
int mfma_conv_stats_finalize(const ConvGeom& g, const float* slab, const float* drop, float eps, float* mean,
                             float* scale, hipStream_t st);
// slab[(y * gx + workgroup) * 4 + wave][n][cb][2] of per-wave (sum, sum of squares) -> mean / scale (or, scale == NULL,
// the two means)
int stats_slab_finalize_launch(const float* slab, int gx, int cb, int N, int C, double invV, const float* drop, float eps,
                               float* mean, float* scale, hipStream_t st);
// the sliding 32-channel kernel can take the InstanceNorm + LeakyReLU backward sums of its output (input-gradient role)
bool mfma_conv_can_fuse_bwd_sums(const ConvGeom& g);
// slab of backward sums -> m12[n][c] = (mean g', mean g' xhat)
int mfma_conv_bwd_sums_finalize(const ConvGeom& g, const float* slab, float* m12, hipStream_t st);
bool mfma_wgrad_eligible(const WgradGeom& g, int dtype);
size_t wgrad_mfma_ws_bytes(const WgradGeom& g);
int wgrad_mfma_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, WgradGeom g,
                      hipStream_t st);

// conv_slide32.hip (3x3x3 stride-1, 32 -> 32 / 64 channels: D-sliding plane ring on v_mfma_f32_16x16x32, a wave = all 32
// couts of a 4 x 16 quarter of the column, the whole weight resident in AGPRs)
struct SlidePlan {
    int dsplit, DL, tiles_h, tiles_w, units, grid, ny;
};
bool slide_conv_plan(int N, int D, int H, int W, int Cin, int Cout, SlidePlan* out);
// bst_act != NULL (input-gradient role): the slab receives the InstanceNorm + LeakyReLU backward sums of the output
// against the activation bst_act (pitch bst_ld) instead of the forward statistics
// x2 / ldx2 / w2: a second tensor on the output grid (32 channels) and a packed 1x1x1 input-gradient weight: y += W2^T x2
int conv_slide_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                      float* stat_slab, hipStream_t st, const void* bst_act = nullptr, int bst_ld = 0, float slope = 0.f,
                      const void* x2 = nullptr, int ldx2 = 0, const void* w2 = nullptr);

// conv_slide64.hip (3x3x3 stride-1, 64 -> 64 / 128 channels: the same design on v_mfma_f32_16x16x32, a wave = 16 couts)
bool slide64_conv_plan(int N, int D, int H, int W, int Cin, int Cout, SlidePlan* out);
// whole-sample form of the deepest level (conv_ws.hip): split-K slices for this geometry (0 = not its shape); the kernel
// writes fp32 partials [slice][voxel][Cout] for conv_ksplit_reduce_kernel
int conv_ws_slices(int N, int D, int H, int W, int Cin, int Cout);
int conv_ws_launch(const void* x, const void* w, float* part, int N, int D, int H, int W, int Cin, int Cout, int ldx,
                   int flip, int slices, hipStream_t st);
int conv_slide64_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                        float* stat_slab, hipStream_t st, const void* bst_act = nullptr, int bst_ld = 0, float slope = 0.f);

// conv_s2.hip (the stride-2 forms of the large levels: LDS-DMA plane ring, producer wave, weights in registers)
bool convt_s2_tile_eligible(const ConvGeom& g);
size_t convt_s2_tile_slab_bytes(const ConvGeom& g);
int convt_s2_tile_slab_geom(const ConvGeom& g, int* gx, int* cb);
int convt_s2_tile_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                         float* stat_slab, const void* x2, int ldx2, const void* w2, hipStream_t st);

bool conv_s2_tile_eligible(const ConvGeom& g);
size_t conv_s2_tile_slab_bytes(const ConvGeom& g);
int conv_s2_tile_slab_geom(const ConvGeom& g, int* gx, int* cb);
int conv_s2_tile_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, float* stat_slab,
                        const void* w2, const float* bias2, void* y2, int ldy2, hipStream_t st);

// fused_skip.hip (decoder ResBlock tail: 1x1x1 skip conv + InstanceNorm apply + sum + LeakyReLU in one pass)
bool skip1x1_fused_eligible(const ru3d_tensor* x, const ru3d_tensor* y2, const ru3d_tensor* out, int dtype);
int skip1x1_fused_launch(const ru3d_tensor* x, const void* w, const float* bias, const ru3d_tensor* y2, const float* mean,
                         const float* scale, const ru3d_tensor* out, float slope, hipStream_t st);

// norm_small.hip (InstanceNorm + LeakyReLU of the small levels).  in_small_mode: 0 = not its shape (norm.hip's three
// launches), 1 = whole-instance kernel (one launch; can sum a conv's split-K slices itself), 2 = two coalesced kernels.
int in_small_mode(const ru3d_tensor* y, bool all_samples, const ru3d_tensor* a = nullptr, const ru3d_tensor* b = nullptr,
                  const ru3d_tensor* c = nullptr, const ru3d_tensor* d = nullptr);
size_t in_small_ws_bytes(const ru3d_tensor* y);
int in_small_fwd_launch(const ru3d_tensor* y, const float* part, int ksplit, const float* bias, const float* drop,
                        float* mean, float* scale, const ru3d_tensor* res, const ru3d_tensor* out, void* ws, float eps,
                        float slope, hipStream_t st);
int in_small_bwd_launch(const ru3d_tensor* gout, const float* part, int ksplit, const ru3d_tensor* outp,
                        const ru3d_tensor* y, const float* mean, const float* scale, const ru3d_tensor* dy,
                        const ru3d_tensor* gpre, void* ws, float slope, int zero_far, float* gpre_sum, hipStream_t st);

// wgrad_slide.hip (3x3x3 stride-1 weight gradient on the large levels: D-sliding plane ring)
struct WgradSlidePlan {
    int dsplit, DL, tiles_h, tiles_w, units, G, pairs;
};
bool wgrad_slide_plan(const WgradGeom& g, WgradSlidePlan* out);
size_t wgrad_slide_ws_bytes(const WgradGeom& g);
int wgrad_slide_launch(const void* x, const void* dy, float* dw, void* ws, const WgradGeom& g, hipStream_t st);
size_t wgrad_slide_pair_ws_bytes(const WgradGeom& g);
int wgrad_slide_pair_launch(const void* x, const void* dy, const void* dy2, int lddy2, float* dw, float* dw2, void* ws,
                            const WgradGeom& g, hipStream_t st);

// wgrad_s2.hip (3x3x3 stride-2 / ConvTranspose weight gradient on the large levels: one parity-split input tile)
bool wgrad_s2_eligible(const WgradGeom& g);
size_t wgrad_s2_ws_bytes(const WgradGeom& g);
int wgrad_s2_launch(const void* x, const void* dy, float* dw, void* ws, const WgradGeom& g, hipStream_t st);
bool wgrad_s2_pair_eligible(const WgradGeom& g);
size_t wgrad_s2_pair_ws_bytes(const WgradGeom& g);
int wgrad_s2_pair_launch(const void* x, const void* dy, const void* dy2, int lddy2, float* dw, float* dw2, void* ws,
                         const WgradGeom& g, hipStream_t st);

// small_convs.hip (1-channel stem, 2..4-channel head)
bool stem_fwd_eligible(const ConvGeom& g, int dtype, int y_dtype, const void* res);
int stem_fwd_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, int dtype,
                    hipStream_t st);
bool head_fwd_eligible(const ConvGeom& g, int dtype, const void* res);
int head_fwd_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, int dtype,
                    int y_dtype, hipStream_t st);
bool head_dgrad_eligible(const ConvGeom& g, int dtype, int y_dtype, const void* res);
int head_dgrad_launch(const void* dy, const void* w, void* dx, const ConvGeom& g, int dtype, hipStream_t st);
bool stem_wgrad_eligible(const WgradGeom& g);
size_t stem_wgrad_ws_bytes(const WgradGeom& g);
int stem_wgrad_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, const WgradGeom& g,
                      int dtype, hipStream_t st, float* db = nullptr);
bool stem_wgrad_gives_bias(const WgradGeom& g, int dtype);
// the head's whole backward in one pass over (a, dlogits fp32): dx, dW, db
bool head_bwd_eligible(int Cin, int Cout, int dtype);
size_t head_bwd_ws_bytes(int64_t P, int Cin);
int head_bwd_launch(const void* a, int lda, const float* dlog, int ldd, const float* w, int cin_real, int Cin, int Cout,
                    void* dx, int lddx, float* dw, float* db, void* ws, int64_t P, hipStream_t st);
bool head_wgrad_eligible(const WgradGeom& g, int dtype);
size_t head_wgrad_ws_bytes(const WgradGeom& g);
int head_wgrad_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, const WgradGeom& g,
                      int dtype, hipStream_t st);

}  // namespace RU3D_NS
using namespace RU3D_NS;
