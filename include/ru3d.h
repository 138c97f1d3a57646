/*
 * ru3d.h - C ABI of the MI355X-native 3D U-Net training hot path (libru3d.so).
 *
 * The reference (icrdr/3D-UNet-Renal-Anatomy-Extraction) has no FFI of its own: its hot path is
 * entered through `nn.Module.__call__` and every arithmetic op is delegated to torch.nn.  This
 * header is the boundary the build introduces *beneath* the reference's Python classes
 * (network.py / loss.py / trainer.py keep their names and signatures); each entry point cites the
 * reference call site whose arithmetic it replaces.
 *
 * Conventions
 *  - plain C only: scalars, raw device pointers, explicit shapes; no C++/torch types.
 *  - the CALLER owns every buffer, including workspaces (sizes from the *_workspace_bytes calls);
 *    the library never allocates or frees device memory and keeps no mutable global state.
 *  - activations are NDHWC ("channels last") in HBM: element (n,d,h,w,c) of a tensor lives at
 *    ptr[(((n*D+d)*H+h)*W+w)*ld + c]; `ld` (>= c) is the voxel pitch in elements, so a tensor can be
 *    a channel slice of a wider buffer (this is how the skip-concat is written in place).
 *  - every call only enqueues work on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream).  No call synchronises the device.
 *  - return value: 0 = OK; < 0 = invalid argument / unsupported shape (see ru3d_last_error());
 *    > 0 = hipError_t from the launch.  Never throws, never exits.
 *  - thread safety: re-entrant; the error string is thread-local (PyTorch runs backward on its own
 *    autograd thread).
 */
#ifndef RU3D_H
#define RU3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RU3D_VERSION 201

/* storage dtypes of activations / packed weights (accumulation is always fp32).  RU3D_F16 is the reference's
 * mixed-precision arithmetic (apex O1: fp16 convolutions with fp32 accumulation, trainer.py:492-493, 538-542) and
 * needs loss scaling on the host side; RU3D_BF16 needs none.  A call uses ONE 16-bit type throughout. */
enum { RU3D_F32 = 0, RU3D_BF16 = 1, RU3D_F16 = 2 };

/* label dtypes accepted by the loss kernels (reference passes int64: loss.py:27) */
enum { RU3D_LABEL_I64 = 0, RU3D_LABEL_U8 = 1 };

/* which reference weight tensor a packed weight is made from, and for which kernel */
enum {
    RU3D_ROLE_CONV_FWD = 0,    /* nn.Conv3d weight [Cout][Cin][k^3]          -> forward          */
    RU3D_ROLE_CONV_DGRAD = 1,  /* nn.Conv3d weight                            -> input gradient   */
    RU3D_ROLE_CONVT_FWD = 2,   /* nn.ConvTranspose3d weight [Cin][Cout][k^3]  -> forward          */
    RU3D_ROLE_CONVT_DGRAD = 3, /* nn.ConvTranspose3d weight                   -> input gradient   */
    RU3D_ROLE_BIAS = 4         /* bias vector [Cout] -> fp32 [Cout padded] (ru3d_pack_weights only)  */
};

/* loss kinds: which reference module the finalize step reproduces */
enum {
    RU3D_LOSS_HYBIRD = 0,   /* loss.py:196-254 HybirdLoss                    */
    RU3D_LOSS_DICELOSS = 1, /* loss.py:123-166 DiceLoss                      */
    RU3D_LOSS_FOCAL = 2,    /* loss.py:169-193 FocalLoss (+ focal_loss :51)  */
    RU3D_LOSS_DICE = 3      /* loss.py:85-120  Dice (metric)                 */
};

typedef struct ru3d_tensor {
    void* ptr;          /* device pointer to element (0,0,0,0,0)        */
    int32_t n, d, h, w; /* batch and spatial extents                     */
    int32_t c;          /* channels                                      */
    int32_t ld;         /* voxel pitch in elements (>= c; split: >= cseg) */
    /* Split ("planar") channel layout, cseg != 0: the channels are c / cseg segments of cseg channels, segment s a dense
     * NDHWC tensor of its own at ptr + s * seg_stride elements (voxel pitch ld): element (n,d,h,w,ch) sits at
     * ptr[(ch / cseg) * seg_stride + (((n*D+d)*H+h)*W+w) * ld + ch % cseg].  This is how torch.cat((up, skip), dim=1)
     * (reference network.py:350) is held on the full-resolution level, where the two 32-channel halves interleaved in
     * one 128-byte row would cost every reader / writer of ONE half double line traffic: both halves stay contiguous
     * tensors and only the kernels that consume the concat (ru3d_planar_concat_supported lists them) take the pair.
     * Every other entry point rejects a split tensor. */
    int32_t cseg;
    int64_t seg_stride;
} ru3d_tensor;

int ru3d_version(void);
/* thread-local, valid until the next failing call on this thread */
const char* ru3d_last_error(void);

/* ------------------------------------------------------------------ weights ----------------- */
/* Number of bytes of the packed form of a weight (layout is private to the library and depends on
 * (cin, cout, k, stride, dtype, role); `stride` is the nn.Module's stride, 2 for the ConvTranspose3d roles). */
size_t ru3d_packed_weight_bytes(int cout, int cin, int k, int stride, int role, int dtype);
/* src: fp32 weight in the reference's layout (state_dict tensor, contiguous).  dst: packed. */
int ru3d_pack_weight(const float* src, void* dst, int cout, int cin, int k, int stride, int role, int dtype,
                     void* stream);

/* Same, for up to RU3D_PACK_MAX weights in ONE launch (a ResBlock's three convs x {forward, dgrad} + biases).
 * Channel padding: widths that are not multiples of 32 (the reference's default num_features = 30,
 * network.py:107, gives 30/60/120/240/480) run on the MFMA kernels with activations padded to the next multiple
 * of 32 and exact zeros in the pad lanes.  cout_seg / cin_seg != 0 ask for the packed weight of such a layer:
 * that dimension of the module's weight consists of (dim / seg) segments of `seg` real channels, each padded with
 * zero rows / columns to the next multiple of 32 (cin_seg = 30 for cin = 60 is the decoder's concat input
 * 30 | 30 -> 32 | 32, network.py:350).  dst is then ru3d_packed_weight_bytes(padded cout, padded cin, ...) long.
 * Role RU3D_ROLE_BIAS pads a bias vector the same way (src fp32 [cout], dst fp32 [padded cout]; k = 1). */
#define RU3D_PACK_MAX 40
typedef struct ru3d_pack_item {
    const float* src; /* fp32 weight, reference layout                  */
    void* dst;        /* packed output, ru3d_packed_weight_bytes() long  */
    int32_t cout, cin, k, stride, role;
    int32_t cout_seg, cin_seg; /* 0 = no padding of that dimension       */
} ru3d_pack_item;
int ru3d_pack_weights(const ru3d_pack_item* items, int count, int dtype, void* stream);

/* Inverse of the channel padding for a weight gradient: src = fp32 [cout_p][cin_p][taps] as the wgrad entry points
 * write it for padded activations, dst = fp32 [cout][cin][taps] (the parameter's shape), segments as above. */
int ru3d_unpad_weight_grad(const float* src, float* dst, int cout, int cin, int taps, int cout_seg, int cin_seg,
                           void* stream);

/* ------------------------------------------------------------------ convolutions ------------ */
/* nn.Conv3d(k in {1,3}, stride in {1,2}, padding=k/2) forward (network.py:394-395,403,541-547).
 * y = conv(x) + bias (+ res).  bias (fp32 [Cout]) and res may be NULL.  y_dtype may be RU3D_F32
 * while x is bf16 (the logits head). */
/* Optional scratch of ru3d_conv3d_fwd / ru3d_conv3d_dgrad (call with (dy, dx) for the latter): split-K partials of the
 * deepest level; 0 for most shapes.  ws may be NULL (the launch then takes a path that needs none). */
size_t ru3d_conv3d_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype);
int ru3d_conv3d_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* res,
                    const ru3d_tensor* y, int k, int stride, int dtype, int y_dtype, void* ws,
                    size_t ws_bytes, void* stream);
/* The same forward conv FOLLOWED by the InstanceNorm statistics of its output (ResBlock: conv -> dropout ->
 * norm, network.py:411-414): y = conv(x) + bias, then mean / scale exactly as ru3d_instnorm_stats(y, ...)
 * would produce.  On the producer/consumer MFMA kernel the sums are accumulated in the conv epilogue (no
 * second pass over y); other shapes run the two kernels back to back.  Workspace from the _bytes query. */
size_t ru3d_conv3d_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype);
int ru3d_conv3d_fwd_in(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y, int k,
                       int stride, int dtype, const float* drop_scale, float* mean, float* scale, void* ws,
                       size_t ws_bytes, float eps, void* stream);
/* conv + InstanceNorm statistics + apply + LeakyReLU behind one entry point - a ResBlock's conv -> dropout -> norm ->
 * nonlin chain (reference network.py:405-416) and, with `res`, its tail lrelu(IN(conv2(x)) + skip):
 *     y = conv(x) + bias;  (mean, scale) as ru3d_conv3d_fwd_in;  out = lrelu((y - mean) * scale (+ res))
 * y is kept (the backward reads it).  On small levels (<= 4096 voxels per sample, 16-bit storage) the statistics, their
 * finalize and the apply are one whole-instance kernel that also sums the conv's split-K slices; elsewhere the call is
 * ru3d_conv3d_fwd_in + ru3d_in_lrelu_fwd.  Workspace: ru3d_conv3d_fwd_in_lrelu_workspace_bytes. */
size_t ru3d_conv3d_fwd_in_lrelu_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype);
int ru3d_conv3d_fwd_in_lrelu(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y, int k,
                             int stride, int dtype, const float* drop_scale, float* mean, float* scale,
                             const ru3d_tensor* res, const ru3d_tensor* out, float slope, void* ws, size_t ws_bytes,
                             float eps, void* stream);
/* 1 when a decoder ResBlock whose input is the concat of two `cseg`-channel tensors on an n x d x h x w grid can take that
 * input as a split tensor through every kernel of its forward and backward: ru3d_conv3d_fwd / _fwd_in / _fwd_in_lrelu (x),
 * ru3d_skip1x1_in_lrelu_fwd (x), ru3d_conv3d_wgrad k = 3 and k = 1 (x), ru3d_conv3d_s1_dgrad_pair (dx).  cout: the
 * block's output channels. */
int ru3d_planar_concat_supported(int n, int d, int h, int w, int cseg, int cout, int dtype);
/* input gradient of the same conv: dx = conv_dgrad(dy) (+ res).  w_packed made with ROLE_CONV_DGRAD. */
int ru3d_conv3d_dgrad(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* res,
                      const ru3d_tensor* dx, int k, int stride, int dtype, void* ws, size_t ws_bytes,
                      void* stream);
/* The input gradient of a decoder ResBlock's two stride-1 convs of the same input in ONE launch (network.py:394 conv1
 * k3 s1 and :403 skip_conv k1 s1, :406-411): dx = conv3_dgrad(dy; w3_packed) + conv1_dgrad(dy2; w1_packed), dy and dy2
 * on the same grid with the same channel count.  `_supported` as above. */
int ru3d_conv3d_s1_dgrad_pair_supported(const ru3d_tensor* dy, const ru3d_tensor* dy2, const ru3d_tensor* dx, int dtype);
int ru3d_conv3d_s1_dgrad_pair(const ru3d_tensor* dy, const void* w3_packed, const ru3d_tensor* dy2, const void* w1_packed,
                              const ru3d_tensor* dx, int dtype, void* stream);
/* The forward of a pooling ResBlock's two stride-2 convs of the same input in ONE launch (network.py:394 conv1 k3 s2 p1,
 * :403 skip_conv k1 s2; :406-411), plus the InstanceNorm statistics of conv1's output as ru3d_conv3d_fwd_in takes them:
 *     y3 = conv3(x; w3, b3), (mean, scale) = IN statistics of Dropout3d(y3), y1 = conv1(x; w1, b1)
 * `_supported`: a fused kernel exists for the shapes; workspace: ru3d_conv3d_s2_pair_fwd_in_workspace_bytes. */
int ru3d_conv3d_s2_pair_fwd_in_supported(const ru3d_tensor* x, const ru3d_tensor* y3, const ru3d_tensor* y1, int dtype);
size_t ru3d_conv3d_s2_pair_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y3, int dtype);
int ru3d_conv3d_s2_pair_fwd_in(const ru3d_tensor* x, const void* w3_packed, const float* b3, const ru3d_tensor* y3,
                               const void* w1_packed, const float* b1, const ru3d_tensor* y1, const float* drop_scale,
                               float* mean, float* scale, void* ws, size_t ws_bytes, float eps, int dtype, void* stream);
/* The input gradient of a pooling ResBlock's two stride-2 convs in ONE launch (reference network.py:394 conv1 k3 s2 p1
 * and :403 skip_conv k1 s2, both applied to the block's input, :406-411):
 *     dx = conv3_dgrad(dy; w3_packed) + conv1_dgrad(dy2; w1_packed) (+ res)
 * dy and dy2 live on the same (pooled) grid with the same channel count.  `_supported` says whether a fused kernel
 * exists for the shapes (the caller otherwise chains two ru3d_conv3d_dgrad calls through `res`). */
int ru3d_conv3d_s2_dgrad_pair_supported(const ru3d_tensor* dy, const ru3d_tensor* dy2, const ru3d_tensor* res,
                                        const ru3d_tensor* dx, int dtype);
int ru3d_conv3d_s2_dgrad_pair(const ru3d_tensor* dy, const void* w3_packed, const ru3d_tensor* dy2,
                              const void* w1_packed, const ru3d_tensor* res, const ru3d_tensor* dx, int dtype,
                              void* stream);
/* weight gradient, written as fp32 in the reference layout [Cout][Cin][k^3] (param.grad). */
size_t ru3d_conv3d_wgrad_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride, int dtype);
int ru3d_conv3d_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, void* ws, size_t ws_bytes,
                      int k, int stride, int dtype, void* stream);

/* A ResBlock's two weight gradients on its input x (reference network.py:403-409: conv1 3x3x3 and skip_conv 1x1x1, both
 * with the block's stride 1 or 2, on the same x) from ONE pass over x:  dw[Cout][Cin][27] = wgrad3(x, dy),
 * dw2[Cout][Cin] = wgrad1(x, dy2).  dy and dy2 have the same shape.  x may be a split tensor (stride 1).  Ask _supported
 * first (the sliding / LDS-DMA weight-gradient kernels' shapes, 16-bit storage); workspace: _workspace_bytes. */
int ru3d_conv3d_wgrad_pair_supported(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, int stride, int dtype);
size_t ru3d_conv3d_wgrad_pair_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, int stride,
                                              int dtype);
int ru3d_conv3d_wgrad_pair(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, float* dw, float* dw2,
                           void* ws, size_t ws_bytes, int stride, int dtype, void* stream);
/* The same weight gradient AND the conv's bias gradient db[co] = sum over samples and voxels of dy (autograd of the bias of
 * network.py:541 conv / any nn.Conv3d) behind one entry point: the stem's MFMA kernel delivers db from its own pass over dy
 * (an extra all-ones row of its im2col operand); other shapes run ru3d_conv3d_wgrad + ru3d_channel_sum. */
size_t ru3d_conv3d_wgrad_bias_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride, int dtype);
int ru3d_conv3d_wgrad_bias(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                           int k, int stride, int dtype, void* stream);
/* The whole backward of the 1x1x1 head conv (network.py:547 `fc`, Cout <= 4) in one pass: x = the head's input (16-bit
 * storage), dlogits = the loss's gradient as fp32 [N][D][H][W][Cout] (it is rounded to the storage type in registers, the
 * value the unfused path's cast stored), weight = the fp32 parameter [Cout][cin_real] (cin_real < x->c: padded channels),
 * -> dx (storage type, pad lanes 0), dw [Cout][cin_real], db [Cout] (may be NULL). */
int ru3d_head_bwd_supported(const ru3d_tensor* x, const ru3d_tensor* dlogits, const ru3d_tensor* dx, int dtype);
size_t ru3d_head_bwd_workspace_bytes(const ru3d_tensor* x, int dtype);
int ru3d_head_bwd(const ru3d_tensor* x, const ru3d_tensor* dlogits, const float* weight, int cin_real, const ru3d_tensor* dx,
                  float* dw, float* db, void* ws, size_t ws_bytes, int dtype, void* stream);

/* nn.ConvTranspose3d(k3,s2,p1) followed by ConstantPad3d((0,1,0,1,0,1),0) (network.py:312-314):
 * y has extents 2*x.{d,h,w}; its far planes are written as exact zeros (no bias there). */
int ru3d_convtranspose3d_k3s2p1_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias,
                                    const ru3d_tensor* y, int dtype, void* stream);
/* The same followed by the InstanceNorm statistics of y (network.py:315; the zero far planes count): mean / scale as
 * ru3d_instnorm_stats writes them; fused into the transposed conv's epilogue where a kernel for the shape exists. */
size_t ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int dtype);
int ru3d_convtranspose3d_k3s2p1_fwd_in(const ru3d_tensor* x, const void* w_packed, const float* bias,
                                       const ru3d_tensor* y, float* mean, float* scale, void* ws, size_t ws_bytes,
                                       float eps, int dtype, void* stream);
/* dy must have zero far planes (ru3d_in_lrelu_bwd(zero_far=1) guarantees it). */
int ru3d_convtranspose3d_k3s2p1_dgrad(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* dx,
                                      int dtype, void* stream);
size_t ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, int dtype);
/* dw fp32 in the reference layout [Cin][Cout][27]. */
int ru3d_convtranspose3d_k3s2p1_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, void* ws,
                                      size_t ws_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------ InstanceNorm + LeakyReLU - */
size_t ru3d_reduce_workspace_bytes(const ru3d_tensor* t);
/* nn.InstanceNorm3d(eps, affine=False) statistics (network.py:401,414), with the preceding
 * nn.Dropout3d folded in: drop_scale[n*C+c] in {0, 1/(1-p)} or NULL.
 * Outputs mean[n*C+c] of the raw y and scale = s / sqrt(s^2 * var + eps), so that
 * x_hat = (y - mean) * scale equals InstanceNorm(Dropout3d(y)). */
int ru3d_instnorm_stats(const ru3d_tensor* y, const float* drop_scale, float* mean, float* scale,
                        void* ws, size_t ws_bytes, float eps, int dtype, void* stream);
/* The tail of a decoder ResBlock in one pass (network.py:403, 406-409, 414-416): out = LeakyReLU((y - mean) * scale +
 * conv1x1(x; w_packed, bias), slope) - the skip conv's output is never stored.  `_supported`: a kernel exists for the
 * shapes (the caller otherwise runs ru3d_conv3d_fwd(k = 1) + ru3d_in_lrelu_fwd(res = skip)). */
int ru3d_skip1x1_in_lrelu_fwd_supported(const ru3d_tensor* x, const ru3d_tensor* y, const ru3d_tensor* out, int dtype);
int ru3d_skip1x1_in_lrelu_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y,
                              const float* mean, const float* scale, const ru3d_tensor* out, float slope, int dtype,
                              void* stream);
/* out = LeakyReLU((y - mean) * scale (+ res), slope)  (network.py:414,416; 315-316). */
int ru3d_in_lrelu_fwd(const ru3d_tensor* y, const float* mean, const float* scale, const ru3d_tensor* res,
                      const ru3d_tensor* out, float slope, int dtype, void* stream);
/* backward of the above: given gout = dL/dout, writes dy = dL/dy and, when the forward had a residual,
 * gpre = gout * LeakyReLU'(out) = dL/dres.  CONTRACT: gpre != NULL <=> the forward was called with a residual.
 * Without a residual out = LeakyReLU(x_hat) is invertible, so x_hat is recovered from `out` and `y` is not
 * read at all (one tensor pass less in each of the two kernels).  zero_far=1 additionally zeroes the last plane
 * of each spatial axis of dy (the constant-pad planes of ConvTrans3D).
 * gpre_sum (optional, C floats): sum over n and voxels of gpre - the bias gradient of a conv that feeds the
 * residual input (skip_conv, network.py:407-409); it falls out of the reduction this call runs anyway.
 * dy_sum (optional, C floats): sum over n and voxels of the stored dy - the bias gradient of the conv or transposed
 * conv that produced y (network.py:290-300); per-block sums in the epilogue of the pass that writes dy. */
int ru3d_in_lrelu_bwd(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y, const float* mean,
                      const float* scale, const ru3d_tensor* dy, const ru3d_tensor* gpre, void* ws,
                      size_t ws_bytes, float slope, int zero_far, float* gpre_sum, float* dy_sum, int dtype,
                      void* stream);
/* out[c] = sum over n,d,h,w of t (bias gradients). */
int ru3d_channel_sum(const ru3d_tensor* t, float* out, void* ws, size_t ws_bytes, int dtype, void* stream);
/* nn.Dropout3d(p) channel mask (network.py:397-398,412-413): scale[n*C+c] = keep ? 1/(1-p) : 0,
 * counter-based RNG keyed by (seed, offset). */
int ru3d_dropout3d_scale(float* scale, int count, float p, uint64_t seed, uint64_t offset, void* stream);
/* The same draw with the counter offset completed ON THE DEVICE: offset + *offset_base.  For a training step captured
 * in a hipGraph - the launch arguments are frozen at capture; the host advances *offset_base between replays, so replay
 * k draws exactly the masks the eager step k would have drawn. */
int ru3d_dropout3d_scale_dev(float* scale, int count, float p, uint64_t seed, uint64_t offset,
                             const uint64_t* offset_base, void* stream);

/* Elementwise pieces of the attention gate (AttBlock, network.py:353-371: x = conv(x); g = conv(gate);
 * rate = sigmoid(conv(lrelu(x + g))); return x * rate) and of its backward; the convolutions are ru3d_conv3d_*.
 *   op 0: o1 = lrelu(a)                                  op 1: o1 = a * sigmoid(b)
 *   op 2: o1 = c * sigmoid(b),  o2 = c * a * sigmoid'(b)  (c = upstream gradient, a = gated tensor, b = pre-sigmoid)
 *   op 3: o1 = c * lrelu'(a),   o2 = o1 + b               (a = the lrelu OUTPUT, c = its gradient, b = another share) */
int ru3d_pointwise(int op, const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* c, const ru3d_tensor* o1,
                   const ru3d_tensor* o2, float slope, int dtype, void* stream);

/* ------------------------------------------------------------------ layout helpers ---------- */
/* dst[...,c] = src[...,c] for c < src.c (torch.cat along channels, network.py:350, written in place). */
int ru3d_copy_channels(const ru3d_tensor* src, const ru3d_tensor* dst, int dtype, void* stream);
/* dst = src + add, same shape (gradient accumulation of the skip branch). */
int ru3d_add(const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* dst, int dtype, void* stream);
/* dst (dtype `dst_dtype`) = src (fp32), same shape: fp32 logits-gradient -> storage dtype. */
int ru3d_cast_f32(const ru3d_tensor* src, const ru3d_tensor* dst, int dst_dtype, void* stream);
/* NCDHW fp32 (reference tensor layout) -> NDHWC `dtype`, and back. */
int ru3d_ncdhw_to_ndhwc(const float* src, const ru3d_tensor* dst, int dtype, void* stream);
int ru3d_ndhwc_to_ncdhw(const ru3d_tensor* src, float* dst, int dtype, void* stream);

/* ------------------------------------------------------------------ loss -------------------- */
/* Fused softmax + focal + Tversky sums (loss.py:7-48, 51-82, 218-254): one pass over logits+labels.
 * logits: fp32, element (n, c, v) at logits[n*stride_n + c*stride_c + v*stride_v] (covers both the
 * NDHWC tensor the network writes and a plain NCDHW tensor).  labels: [N*V] int64 or uint8.
 * state: caller-owned device buffer of ru3d_loss_state_bytes(C) bytes; receives the class sums, the
 * backward coefficients and the number of out-of-range labels.  loss_out: 1 fp32 on device.
 * C == 1 uses sigmoid with an all-ones one-hot (what F.one_hot(target, 1) yields for valid targets). */
size_t ru3d_loss_state_bytes(int num_classes);
/* byte offset, inside the state buffer, of the int32 count of labels outside [0, C) that the forward pass found (the
 * reference's F.one_hot raises for them, loss.py:27): a host can read these 4 bytes at its next read-back. */
size_t ru3d_loss_state_bad_labels_offset(void);
size_t ru3d_loss_workspace_bytes(int n, int64_t v, int num_classes);
int ru3d_loss_fwd(const float* logits, int64_t stride_n, int64_t stride_c, int64_t stride_v, const void* labels,
                  int label_dtype, int n, int64_t v, int num_classes, int kind, float gamma, const float* weight_v,
                  float alpha, float beta, float smooth, void* state, float* loss_out, void* ws, size_t ws_bytes,
                  void* stream);
/* dlogits (same strides as logits; dtype f32 or bf16) = grad_out[0] * dLoss/dlogits. */
int ru3d_loss_bwd(const float* logits, int64_t stride_n, int64_t stride_c, int64_t stride_v, const void* labels,
                  int label_dtype, int n, int64_t v, int num_classes, float gamma, const void* state,
                  const float* grad_out, void* dlogits, int dlogits_dtype, void* stream);
/* functional `dice` of loss.py:32-48 on two flat fp32 vectors (used by trainer.evaluate_case). */
int ru3d_tversky(const float* p, const float* g, int64_t count, float alpha, float beta, float smooth,
                 float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ sliding-window inference */
/* predict_per_patch (trainer.py:17-98).  `acc` is a zero-initialised [X, Y, Z, C] fp32 volume, `cnt` a
 * zero-initialised [X, Y, Z] fp32 volume (the reference's `result` / `result_n`, trainer.py:42-43).
 * accumulate: acc[window] += softmax_c(logits[sample]) (sigmoid when C == 1), cnt[window] += 1, the window
 * being [ox, ox+d) x [oy, oy+h) x [oz, oz+w) (trainer.py:72-80).  Launches for overlapping windows must be
 * issued on one stream; the sum order is then the launch order, as in the reference's patch loop. */
int ru3d_predict_accumulate(const ru3d_tensor* logits, int dtype, int sample, float* acc, float* cnt, int X, int Y,
                            int Z, int ox, int oy, int oz, void* stream);
/* merge (trainer.py:85-98) over the crop [cx, cx+sx) x [cy, cy+sy) x [cz, cz+sz) of the padded volume:
 * one_hot != 0: out = float [sx, sy, sz, C] = acc / cnt (NaN where no window reached, as in the reference);
 * one_hot == 0: out = uint8 [sx, sy, sz]: C == 1 -> round(acc / cnt); C > 1 -> argmax_c softmax_c(acc / cnt)
 * (the reference's second softmax); uncovered voxels -> 0. */
int ru3d_predict_merge(const float* acc, const float* cnt, int X, int Y, int Z, int num_classes, int cx, int cy,
                       int cz, int sx, int sy, int sz, int one_hot, void* out, void* stream);

/* ------------------------------------------------------------------ patch sampling + augmentation */
/* The reference's training transform chain on the device: RandomRescaleCrop -> RandomMirror ->
 * RandomContrast -> RandomBrightness -> RandomGamma -> ToTensor (transform.py:573-652, 279-301, 176-259, 156-163;
 * nb_train_iia.py:30-39).  The HOST makes the random draws (in the reference's order) and passes them here; the
 * kernels are deterministic.  image: fp32 volume [X][Y][Z][C] (the reference's channels-last case layout), label:
 * [X][Y][Z] uint8 or int64.  out_image: fp32 [C][px][py][pz] (ToTensor layout = one NCDHW sample), out_label: int64
 * [px][py][pz].  Resampling restates scipy.ndimage.zoom(order=1): corner-aligned linear interpolation in float64. */
typedef struct ru3d_patch_params {
    int32_t lo[3];      /* crop box lower corner in the volume, per axis (may lie outside: constant padding)      */
    int32_t before[3];  /* crop box size = round(patch / scale), transform.py:617                                  */
    int32_t patch[3];   /* output patch size                                                                       */
    int32_t flip[3];    /* RandomMirror: reverse this output axis                                                   */
    float image_cval;   /* image_pad_cval                                                                           */
    int32_t label_cval; /* label_pad_cval                                                                           */
    int32_t do_contrast, do_brightness, do_gamma;
    float contrast, brightness, gamma; /* the drawn factors                                                         */
    float gamma_eps;    /* adjust_gamma's epsilon (1e-7), transform.py:188                                          */
} ru3d_patch_params;
size_t ru3d_augment_workspace_bytes(int px, int py, int pz);
/* mask (1 device uint32): bit min(v, 31) set for every label value v inside the crop box (pad value included). */
int ru3d_augment_label_presence(const void* label, int label_dtype, int X, int Y, int Z, const int32_t* lo,
                                const int32_t* before, int label_cval, uint32_t* mask, void* stream);
/* label / out_label may both be NULL (image only), or image / out_image (label only).  presence_mask: device uint32 from the call above (the label rule
 * of transform.py:47-48 needs the number of classes in the crop); NULL = treat as >= 3 classes. */
int ru3d_augment_patch(const float* image, const void* label, int label_dtype, int X, int Y, int Z, int C,
                       const ru3d_patch_params* p, const uint32_t* presence_mask, float* out_image, int64_t* out_label,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ optimizer --------------- */
/* torch.optim.Adam step (nb_train_iia.py:18 defaults), fused over one flat fp32 parameter run.
 * grad may be bf16/f32 (grad_dtype); bias corrections are passed in by the host. */
int ru3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                   float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                   float grad_scale, void* stream);

/* The same update for MANY parameter tensors in one launch.  `tensors` and `block_map` are DEVICE arrays:
 * block b of the grid updates elements [chunk*chunk_elems, +chunk_elems) of tensors[block_map[2b]] with
 * chunk = block_map[2b+1].  Tensors whose grad pointer is NULL are skipped (no gradient this step). */
typedef struct ru3d_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t count;
} ru3d_adam_tensor;
int ru3d_adam_multi(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                    float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                    float grad_scale, void* stream);

/* ru3d_adam_multi with the per-step scalars read from device memory - hyper[8] = {lr, beta1, beta2, eps, bias_corr1,
 * bias_corr2, grad_scale, sqrtf(bias_corr2)} - so that a launch captured in a hipGraph follows the step count and the learning-rate
 * schedule: the host rewrites the 32 bytes before each replay. */
int ru3d_adam_multi_dev(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                        const float* hyper, void* stream);

/* The loss scaler of fp16 training held on the DEVICE (32 bytes), so that a whole fp16 step - scaled loss, backward,
 * overflow check, Adam, scaler update - is a fixed launch sequence a hipGraph can replay (the reference trains with apex
 * O1: trainer.py:492-493, 538-542).  ru3d_grad_scale_check(scale = 1) writes found_inf; ru3d_adam_multi_amp is
 * ru3d_adam_multi_dev that skips itself on found_inf != 0, multiplies the gradients by inv_scale and takes the Adam step
 * number from hyper[5] (an int32 in the float slot: steps taken before the capture) + steps + 1, with hyper[4] / hyper[7]
 * the residuals beta - (float)beta of the two betas (bias corrections in double); ru3d_amp_update then
 * applies apex's schedule (overflow: scale *= backoff, tracker = 0, skipped++; clean: steps++, tracker++, after
 * growth_interval clean steps scale *= growth) and clears found_inf.  The host reads the block back when it wants to
 * know (state_dict, logging), not every step. */
typedef struct ru3d_amp_state {
    float scale, inv_scale, found_inf;
    int32_t tracker, skipped, steps, reserved0, reserved1;
} ru3d_amp_state;
int ru3d_adam_multi_amp(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                        const float* hyper, const ru3d_amp_state* amp, void* stream);
int ru3d_amp_update(ru3d_amp_state* amp, float growth_factor, float backoff_factor, int growth_interval, float min_scale,
                    float max_scale, void* stream);

/* conv3d input gradient FOLLOWED by the InstanceNorm + LeakyReLU backward of the tensor it differentiates (ResBlock
 * backward, network.py:411-416 read backwards: da = conv2^T(dy); dyn = d/dy1 of lrelu(IN(y1)) given da) - the
 * mirror image of ru3d_conv3d_fwd_in.  `act` = lrelu(IN(y1)) as the forward produced it, mean / scale = that
 * InstanceNorm's statistics; da receives the conv's result (gradient wrt act), dyn the gradient wrt y1.  On the
 * shapes of the sliding 32-channel kernel the two sums the backward needs (sum g', sum g' xhat) are taken in the
 * conv's epilogue and the separate reduction pass over (da, act) is not run; elsewhere the call is
 * ru3d_conv3d_dgrad + ru3d_in_lrelu_bwd.  Workspace: ru3d_conv3d_dgrad_in_bwd_workspace_bytes. */
size_t ru3d_conv3d_dgrad_in_bwd_workspace_bytes(const ru3d_tensor* dy, const ru3d_tensor* da, int k, int stride,
                                                int dtype);
int ru3d_conv3d_dgrad_in_bwd(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* act, const float* mean,
                             const float* scale, const ru3d_tensor* da, const ru3d_tensor* dyn, int k, int stride,
                             float slope, int dtype, void* ws, size_t ws_bytes, void* stream);
/* The apply pass of ru3d_in_lrelu_bwd alone (no residual form), with m12[n][c] = (mean g', mean g' xhat) given. */
int ru3d_in_lrelu_bwd_apply(const ru3d_tensor* gout, const ru3d_tensor* out, const float* mean, const float* scale,
                            const float* m12, const ru3d_tensor* dy, float slope, int zero_far, int dtype, void* stream);

/* ------------------------------------------------------------------ BatchNorm3d, training mode - */
/* Blocks built with norm_op=nn.BatchNorm3d (network.py:38-69 ResAttrBNUnet3D; nn.BatchNorm3d(C) defaults: affine,
 * running statistics, momentum 0.1, eps 1e-5) in training mode: statistics pooled over the batch, with the preceding
 * nn.Dropout3d's per-(n, c) factor d folded in (network.py:411-413).  Each direction is two calls with the pooled sums
 * handed back in between as doubles, so that a multi-GPU caller can all-reduce them (SyncBN; count is then the global
 * element count N * D * H * W summed over ranks).
 *   stats_pool:     pooled[2c] = sum_n d sum_v y, pooled[2c+1] = sum_n d^2 sum_v y^2     (ws: ru3d_reduce_workspace_bytes)
 *   stats_finalize: mu = pooled[2c] / count, var = pooled[2c+1] / count - mu^2 (biased), r = 1 / sqrt(var + eps);
 *                   a[n][c] = d r, b[n][c] = -mu r (x_hat = y a + b), fscale = gamma a, fshift = beta + gamma b;
 *                   running_mean / running_var (NULL: not tracked) move by `momentum` towards mu / the unbiased var.
 *                   c_real <= c: channels beyond it are zero padding (gamma = 1, beta = 0, no running update).
 *   affine_lrelu_fwd: out = LeakyReLU(y * scale[n][c] + shift[n][c] (+ res), slope)
 *   bwd_pool:       gpre = gout * LeakyReLU'(out) is stored; pooled[2c] = sum gpre (= dbeta of this rank),
 *                   pooled[2c+1] = sum gpre x_hat (= dgamma of this rank)
 *   bwd_apply:      dy = fscale * (gpre - pooled[2c] / count - x_hat * pooled[2c+1] / count); zero_far as in
 *                   ru3d_in_lrelu_bwd; ws: 2 * N * C floats. */
int ru3d_batchnorm_stats_pool(const ru3d_tensor* y, const float* drop_scale, double* pooled, void* ws, size_t ws_bytes,
                              int dtype, void* stream);
int ru3d_batchnorm_stats_finalize(const double* pooled, int n, int c, int c_real, double count, const float* drop_scale,
                                  const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, float* fscale, float* fshift, float* a, float* b, void* stream);
int ru3d_affine_lrelu_fwd(const ru3d_tensor* y, const float* scale, const float* shift, const ru3d_tensor* res,
                          const ru3d_tensor* out, float slope, int dtype, void* stream);
int ru3d_batchnorm_bwd_pool(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y, const float* a,
                            const float* b, const ru3d_tensor* gpre, double* pooled, void* ws, size_t ws_bytes,
                            float slope, int dtype, void* stream);
int ru3d_batchnorm_bwd_apply(const ru3d_tensor* gpre, const ru3d_tensor* y, const float* a, const float* b,
                             const float* fscale, const double* pooled, double count, const ru3d_tensor* dy, void* ws,
                             size_t ws_bytes, int zero_far, int dtype, void* stream);

/* Dynamic loss scaling of the fp16 mode (the reference's apex O1: trainer.py:492-493 amp.scale_loss, 538-542
 * amp.initialize): every gradient named by the table (same layout as ru3d_adam_multi; only grad / count are read) is
 * checked for inf / nan - *found_inf is set to 1.0 when one is found; the caller zeroes it beforehand - and multiplied
 * in place by `scale` (pass 1 / loss_scale; scale == 1 checks without writing). */
int ru3d_grad_scale_check(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                          float scale, float* found_inf, void* stream);

/* ------------------------------------------------------------------ gradient exchange (RCCL) */
/* Data-parallel training: one process per GPU, one exchange per step - the mean of the parameter gradients over
 * ranks.  The reference has no distributed code (it pins one device, nb_train_iia.py:17); these entry points are
 * the build's own contract.  RCCL is resolved with dlopen at first use.
 *   rank 0:      ru3d_comm_unique_id(blob)      -> the host ships the 128-byte blob to every rank (any channel)
 *   every rank:  ru3d_comm_init(&comm, blob, world, rank, device)     (collective: all ranks must call it)
 *   per bucket:  ru3d_comm_allreduce(comm, buf, count, dtype, average, stream)   in place, enqueue-only
 *   at exit:     ru3d_comm_destroy(comm)
 * The communicator handle is the only state the library keeps, and it is owned by the caller. */
#define RU3D_COMM_ID_BYTES 128
int ru3d_comm_unique_id(void* id_out);
int ru3d_comm_init(void** comm_out, const void* unique_id, int world, int rank, int device);
/* buf: device buffer of `count` elements (RU3D_F32 or RU3D_BF16); average != 0 divides the sum by the world size. */
int ru3d_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream);
/* The same exchange as its two halves, each rank owning count_per_rank elements of the bucket (with 7
 * point-to-point xGMI links per GPU a ring all-reduce is bound by one link; reduce-scatter + all-gather lets RCCL use
 * its direct / mesh schedules and leaves room to run the optimizer on the owned shard between the two).  Both in place
 * on a bucket of world * count_per_rank elements: rank r's shard is elements [r * count_per_rank, (r + 1) * ..). */
int ru3d_comm_reduce_scatter(void* comm, void* buf, int64_t count_per_rank, int dtype, int average, void* stream);
int ru3d_comm_all_gather(void* comm, void* buf, int64_t count_per_rank, int dtype, void* stream);
/* 1 when librccl can be loaded in this process (no communicator is created, nothing collective happens): lets every
 * rank vote on the transport BEFORE any rank enters the collective ru3d_comm_init. */
int ru3d_comm_available(void);
int ru3d_comm_destroy(void* comm);
/* Kernel probe (bench.py's `roofline` object): between begin and end the library records a HIP-event pair on the launch
 * stream around the main kernel of every 3x3x3 stride-1 conv launch (forward or input gradient, any entry point) whose
 * output grid is n x d x h x w with cin -> cout channels; `end` waits for them and returns their number and the sum of
 * their durations.  Launches made while the stream is being captured into a graph are skipped. */
int ru3d_probe_begin(int n, int d, int h, int w, int cin, int cout);
int ru3d_probe_end(int* launches, double* total_ms);
/* Compute units the persistent conv kernels launched on a device may occupy (their grids are sized from it); 0 restores
 * the device's count.  A data-parallel rank leaves a few CUs to RCCL's reduction kernels on the side stream.  The value
 * belongs to the DEVICE (the CUs being divided are that device's): every entry point uses the budget of the device its
 * stream lives on.  ru3d_set_cu_budget / ru3d_get_cu_budget address the calling thread's current device. */
int ru3d_set_cu_budget_device(int device, int cus);
int ru3d_set_cu_budget(int cus);
int ru3d_get_cu_budget(void);
/* dst[i] = (dst_dtype)(src[i] * scale) on flat 16-byte-aligned device arrays, f32 <-> bf16: the copy-in / copy-out
 * of a bf16 gradient bucket (half the bytes over xGMI). */
int ru3d_flat_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t count, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RU3D_H */
