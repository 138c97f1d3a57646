// C-ABI entry points for the convolution family + error plumbing.  Validates shapes on the host
// before anything is launched (a faulting kernel can reset the whole GPU node), builds the
// data-movement geometry and dispatches to the MFMA implicit-GEMM kernels (bf16, MFMA-friendly
// channel counts) or to the generic direct kernels (fp32 parity mode, odd channel counts).
#include "common.h"
#include "conv.h"
#include <string>

#ifndef RU3D_STORAGE_F16   // error plumbing and version: defined once, by the bf16 build

static thread_local std::string g_last_error;

int ru3d_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int ru3d_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ru3d_fail((int)e, "%s: %s", what, hipGetErrorString(e));
        return (int)e > 0 ? (int)e : 1;
    }
    return 0;
}

extern "C" int ru3d_version(void) { return RU3D_VERSION; }
extern "C" const char* ru3d_last_error(void) { return g_last_error.c_str(); }
#endif

namespace RU3D_NS {

static ConvGeom fwd_geom(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride);
static bool dtype_ok(int d) { return d == RU3D_F32 || d == RU3D_BF16; }   // fp32 or this build's 16-bit type
static int conv_out(int in, int k, int s) { return (in + 2 * (k / 2) - k) / s + 1; }

// channels (in', out') of the data-movement kernel that consumes a packed weight
static void role_channels(int cout, int cin, int role, int* kin, int* kout, int64_t* s_o, int64_t* s_i, int taps) {
    switch (role) {
        case RU3D_ROLE_CONV_FWD:  // W[co][ci][tap]
            *kin = cin; *kout = cout; *s_o = (int64_t)cin * taps; *s_i = taps; break;
        case RU3D_ROLE_CONV_DGRAD:  // in' = co, out' = ci
            *kin = cout; *kout = cin; *s_o = taps; *s_i = (int64_t)cin * taps; break;
        case RU3D_ROLE_CONVT_FWD:  // W[ci][co][tap], in' = ci, out' = co
            *kin = cin; *kout = cout; *s_o = taps; *s_i = (int64_t)cout * taps; break;
        default:  // CONVT_DGRAD: in' = co, out' = ci
            *kin = cout; *kout = cin; *s_o = (int64_t)cout * taps; *s_i = taps; break;
    }
}

// does the kernel that will consume this packed weight run on the MFMA path?  (all forms; channel counts decide)
static bool role_uses_mfma(int kin, int kout, int k, int stride, int role, int dtype) {
    (void)stride;
    (void)role;
    return mfma_conv_eligible(kin, kout, k, dtype, dtype);
}

extern "C" size_t ru3d_packed_weight_bytes(int cout, int cin, int k, int stride, int role, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_packed_weight_bytes_f16(cout, cin, k, stride, role, dtype));
    if (cout <= 0 || cin <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2) || role < 0 || role > 3 ||
        !dtype_ok(dtype))
        return 0;
    const int taps = k * k * k;
    int kin, kout;
    int64_t s_o, s_i;
    role_channels(cout, cin, role, &kin, &kout, &s_o, &s_i, taps);
    if (role_uses_mfma(kin, kout, k, stride, role, dtype)) return mfma_packed_bytes(kin, kout, taps);
    return (size_t)taps * kin * generic_cout_pad(kout) * (dtype == RU3D_F32 ? 4 : 2);
}

extern "C" int ru3d_pack_weight(const float* src, void* dst, int cout, int cin, int k, int stride, int role, int dtype,
                                void* stream) {
    RU3D_FWD_F16(dtype, ru3d_pack_weight_f16(src, dst, cout, cin, k, stride, role, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(src && dst, "pack_weight: null pointer");
    RU3D_REQUIRE(cout > 0 && cin > 0 && (k == 1 || k == 3) && (stride == 1 || stride == 2),
                 "pack_weight: bad shape cout=%d cin=%d k=%d stride=%d", cout, cin, k, stride);
    RU3D_REQUIRE(role >= 0 && role <= 3 && dtype_ok(dtype), "pack_weight: bad role/dtype");
    const int taps = k * k * k;
    int kin, kout;
    int64_t s_o, s_i;
    role_channels(cout, cin, role, &kin, &kout, &s_o, &s_i, taps);
    if (role_uses_mfma(kin, kout, k, stride, role, dtype))
        return pack_mfma_launch(src, dst, kin, kout, taps, s_o, s_i, 0, as_stream(stream));
    return pack_generic_launch(src, dst, kin, kout, taps, s_o, s_i, 0, dtype, as_stream(stream));
}

static int pad32(int c) { return (c + 31) / 32 * 32; }
// packed extent of a module dimension of `real` channels made of segments of `seg` (0: no padding)
static int padded_dim(int real, int seg) { return seg > 0 ? (real / seg) * pad32(seg) : real; }

extern "C" int ru3d_pack_weights(const ru3d_pack_item* items, int count, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_pack_weights_f16(items, count, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(items && count > 0 && count <= RU3D_PACK_MAX, "pack_weights: count must be 1..%d", RU3D_PACK_MAX);
    RU3D_REQUIRE(dtype_ok(dtype), "pack_weights: bad dtype");
    PackBatch b;
    b.count = count;
    for (int i = 0; i < count; i++) {
        const ru3d_pack_item& it = items[i];
        RU3D_REQUIRE(it.src && it.dst && it.cout > 0 && it.cin > 0 && (it.k == 1 || it.k == 3) &&
                         (it.stride == 1 || it.stride == 2) && it.role >= 0 && it.role <= RU3D_ROLE_BIAS,
                     "pack_weights: bad item %d", i);
        RU3D_REQUIRE(it.cout_seg >= 0 && it.cin_seg >= 0 && (it.cout_seg == 0 || it.cout % it.cout_seg == 0) &&
                         (it.cin_seg == 0 || it.cin % it.cin_seg == 0),
                     "pack_weights: item %d: segments (%d, %d) do not divide (%d, %d)", i, it.cout_seg, it.cin_seg,
                     it.cout, it.cin);
        PackOne& p = b.item[i];
        p.src = it.src;
        p.dst = it.dst;
        if (it.role == RU3D_ROLE_BIAS) {
            p.mfma = 2;
            p.cout = padded_dim(it.cout, it.cout_seg);
            p.cin = 1; p.taps = 1; p.cout_pad = p.cout; p.s_o = 1; p.s_i = 1;
            p.co_real = it.cout_seg; p.co_pad = it.cout_seg ? pad32(it.cout_seg) : 0;
            p.ci_real = 0; p.ci_pad = 0;
            p.total = p.cout;
            continue;
        }
        const int taps = it.k * it.k * it.k;
        int kin, kout;
        int64_t s_o, s_i;
        role_channels(it.cout, it.cin, it.role, &kin, &kout, &s_o, &s_i, taps);   // strides: of the REAL source
        // which module dimension is the kernel's input / output channel dimension
        const bool in_is_cin = (it.role == RU3D_ROLE_CONV_FWD || it.role == RU3D_ROLE_CONVT_FWD);
        const int kin_seg = in_is_cin ? it.cin_seg : it.cout_seg, kout_seg = in_is_cin ? it.cout_seg : it.cin_seg;
        const int kin_p = padded_dim(kin, kin_seg), kout_p = padded_dim(kout, kout_seg);
        p.cin = kin_p; p.cout = kout_p; p.taps = taps; p.s_o = s_o; p.s_i = s_i;
        p.ci_real = kin_seg; p.ci_pad = kin_seg ? pad32(kin_seg) : 0;
        p.co_real = kout_seg; p.co_pad = kout_seg ? pad32(kout_seg) : 0;
        p.mfma = role_uses_mfma(kin_p, kout_p, it.k, it.stride, it.role, dtype) ? 1 : 0;
        p.cout_pad = generic_cout_pad(kout_p);
        p.total = p.mfma ? (int64_t)taps * kin_p * kout_p : (int64_t)taps * kin_p * p.cout_pad;
    }
    // MFMA forms whose packed extents are multiples of 32 go through the fused kernel: one read of the source serves the
    // forward and the input-gradient role of the same weight (the usual request: a ResBlock packs both for each conv)
    PackPairBatch pairs;
    pairs.count = 0;
    if (dtype == RU3D_BF16) {
        for (int i = 0; i < count; i++) {
            PackOne& p = b.item[i];
            const ru3d_pack_item& it = items[i];
            if (p.mfma != 1 || (p.cin % 32) || (p.cout % 32) || (p.taps != 27 && p.taps != 1)) continue;
            // source [a][b][tap]: Conv3d a = cout, ConvTranspose3d a = cin; a-major form = the role whose kout is a
            const bool is_t = it.role == RU3D_ROLE_CONVT_FWD || it.role == RU3D_ROLE_CONVT_DGRAD;
            const bool a_major = it.role == RU3D_ROLE_CONV_FWD || it.role == RU3D_ROLE_CONVT_DGRAD;
            const int a_seg = is_t ? it.cin_seg : it.cout_seg, b_seg = is_t ? it.cout_seg : it.cin_seg;
            const int a_realdim = is_t ? it.cin : it.cout, b_realdim = is_t ? it.cout : it.cin;
            int slot = -1;
            for (int j = 0; j < pairs.count; j++)
                if (pairs.item[j].src == it.src && pairs.item[j].taps == p.taps &&
                    pairs.item[j].adim == padded_dim(a_realdim, a_seg) && pairs.item[j].bdim == padded_dim(b_realdim, b_seg) &&
                    pairs.item[j].a_real == a_seg && pairs.item[j].b_real == b_seg &&
                    (a_major ? !pairs.item[j].dst_a : !pairs.item[j].dst_b))
                    slot = j;
            if (slot < 0) {
                slot = pairs.count++;
                PackPair& q = pairs.item[slot];
                q.src = it.src; q.dst_a = nullptr; q.dst_b = nullptr;
                q.adim = padded_dim(a_realdim, a_seg); q.bdim = padded_dim(b_realdim, b_seg); q.taps = p.taps;
                q.a_real = a_seg; q.a_pad = a_seg ? pad32(a_seg) : 0;
                q.b_real = b_seg; q.b_pad = b_seg ? pad32(b_seg) : 0;
                q.s_a = (int64_t)b_realdim * p.taps;
            }
            (a_major ? pairs.item[slot].dst_a : pairs.item[slot].dst_b) = p.dst;
            p.mfma = 3;   // handled by the fused kernel: pack_batch_kernel skips it
        }
    }
    bool rest = false;
    for (int i = 0; i < count; i++) rest = rest || b.item[i].mfma != 3;
    if (pairs.count) {
        int rc = pack_pair_launch(pairs, as_stream(stream));
        if (rc || !rest) return rc;
    }
    return pack_batch_launch(b, dtype, as_stream(stream));
}

#ifndef RU3D_STORAGE_F16
extern "C" int ru3d_unpad_weight_grad(const float* src, float* dst, int cout, int cin, int taps, int cout_seg,
                                      int cin_seg, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(src && dst && cout > 0 && cin > 0 && taps > 0, "unpad_weight_grad: bad argument");
    RU3D_REQUIRE(cout_seg >= 0 && cin_seg >= 0 && (cout_seg == 0 || cout % cout_seg == 0) &&
                     (cin_seg == 0 || cin % cin_seg == 0), "unpad_weight_grad: bad segments");
    return unpad_weight_launch(src, dst, cout, cin, taps, cout_seg, cout_seg ? pad32(cout_seg) : 0, cin_seg,
                               cin_seg ? pad32(cin_seg) : 0, padded_dim(cin, cin_seg), as_stream(stream));
}
#endif

static int run_conv(const ru3d_tensor* x, const void* w, const float* bias, const ru3d_tensor* res,
                    const ru3d_tensor* y, int k, int stride, int transposed, int flip, int zero_far, int dtype,
                    int y_dtype, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
    ConvGeom g;
    g.N = x->n;
    g.Di = x->d; g.Hi = x->h; g.Wi = x->w; g.Cin = x->c; g.ldx = x->ld;
    g.Do = y->d; g.Ho = y->h; g.Wo = y->w; g.Cout = y->c; g.ldy = y->ld;
    g.CoutPad = generic_cout_pad(y->c);
    g.ldr = res ? res->ld : 0;
    g.k = k; g.stride = stride; g.pad = k / 2;
    g.transposed = transposed; g.zero_far = zero_far; g.flip = flip;
    g.x_cseg = x->cseg; g.x_segstride = x->seg_stride; g.y_cseg = y->cseg; g.y_segstride = y->seg_stride;
    if (g.x_cseg || g.y_cseg) {      // the planar concat: only the MFMA conv kernels of a decoder ResBlock take it
        if (!(mfma_conv_eligible(g.Cin, g.Cout, k, dtype, y_dtype) && mfma_conv_geometry_ok(g)))
            return ru3d_fail(-1, "conv3d: no kernel takes a split tensor for this shape / dtype");
        return conv_mfma_launch(x->ptr, w, bias, res ? res->ptr : nullptr, y->ptr, g, st, nullptr, ws, ws_bytes);
    }
    if (stem_fwd_eligible(g, dtype, y_dtype, res)) return stem_fwd_launch(x->ptr, w, bias, y->ptr, g, dtype, st);
    if (head_fwd_eligible(g, dtype, res)) return head_fwd_launch(x->ptr, w, bias, y->ptr, g, dtype, y_dtype, st);
    if (!bias && head_dgrad_eligible(g, dtype, y_dtype, res)) return head_dgrad_launch(x->ptr, w, y->ptr, g, dtype, st);
    if (mfma_conv_eligible(g.Cin, g.Cout, k, dtype, y_dtype) && mfma_conv_geometry_ok(g))
        return conv_mfma_launch(x->ptr, w, bias, res ? res->ptr : nullptr, y->ptr, g, st, nullptr, ws, ws_bytes);
    // ru3d_pack_weight(s) chose the MFMA fragment layout from (channels, k, dtype) alone: a conv of such a shape that
    // cannot take the MFMA path (16-bit input with an fp32 output other than the small head) must not hand that pack to
    // the direct kernel, which expects [tap][cin][cout_pad]
    if (mfma_conv_eligible(g.Cin, g.Cout, k, dtype, dtype))
        return ru3d_fail(-1, "conv3d: %d -> %d channels k=%d is packed in MFMA fragment order, which has no kernel for "
                             "input dtype %d -> output dtype %d", g.Cin, g.Cout, k, dtype, y_dtype);
    return conv_generic_launch(x->ptr, w, bias, res ? res->ptr : nullptr, y->ptr, g, dtype, y_dtype, st);
}

static bool res_ok(const ru3d_tensor* res, const ru3d_tensor* y) {
    return !res || (tensor_ok(res) && res->n == y->n && res->d == y->d && res->h == y->h && res->w == y->w &&
                    res->c == y->c);
}

// Optional scratch of a conv launch (split-K partials of the deepest level); 0 for most shapes.  The same figure serves
// the input gradient of a stride-1 conv (call it with (dy, dx)).
extern "C" size_t ru3d_conv3d_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_workspace_bytes_f16(x, y, k, stride, dtype));
    if (!tensor_ok_split(x) || !tensor_ok(y) || dtype != RU3D_BF16 || k != 3 || stride != 1) return 0;
    if (!mfma_conv_eligible(x->c, y->c, k, dtype, dtype)) return 0;
    ConvGeom g;
    g.N = x->n;
    g.Di = x->d; g.Hi = x->h; g.Wi = x->w; g.Cin = x->c; g.ldx = x->ld;
    g.Do = y->d; g.Ho = y->h; g.Wo = y->w; g.Cout = y->c; g.ldy = y->ld;
    g.CoutPad = generic_cout_pad(y->c);
    g.ldr = 0;
    g.k = k; g.stride = stride; g.pad = k / 2;
    g.transposed = 0; g.zero_far = 0; g.flip = 0;
    return conv_mfma_ws_bytes(g);
}

extern "C" int ru3d_conv3d_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* res,
                               const ru3d_tensor* y, int k, int stride, int dtype, int y_dtype, void* ws,
                               size_t ws_bytes, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_fwd_f16(x, w_packed, bias, res, y, k, stride, dtype, y_dtype, ws, ws_bytes, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok_split(x) && tensor_ok(y) && w_packed, "conv3d_fwd: bad tensor/weight");
    RU3D_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "conv3d_fwd: k=%d stride=%d unsupported", k, stride);
    RU3D_REQUIRE(dtype_ok(dtype) && dtype_ok(y_dtype) && !(dtype == RU3D_F32 && y_dtype == RU3D_BF16),
                 "conv3d_fwd: bad dtypes %d -> %d", dtype, y_dtype);
    RU3D_REQUIRE(x->n == y->n && y->d == conv_out(x->d, k, stride) && y->h == conv_out(x->h, k, stride) &&
                     y->w == conv_out(x->w, k, stride),
                 "conv3d_fwd: output extents (%d,%d,%d) do not match input (%d,%d,%d) k=%d s=%d", y->d, y->h, y->w,
                 x->d, x->h, x->w, k, stride);
    RU3D_REQUIRE(res_ok(res, y), "conv3d_fwd: residual shape mismatch");
    return run_conv(x, w_packed, bias, res, y, k, stride, 0, 0, 0, dtype, y_dtype, as_stream(stream), ws, ws_bytes);
}

static ConvGeom fwd_geom(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride) {
    ConvGeom g;
    g.N = x->n;
    g.Di = x->d; g.Hi = x->h; g.Wi = x->w; g.Cin = x->c; g.ldx = x->ld;
    g.Do = y->d; g.Ho = y->h; g.Wo = y->w; g.Cout = y->c; g.ldy = y->ld;
    g.CoutPad = generic_cout_pad(y->c);
    g.ldr = 0;
    g.k = k; g.stride = stride; g.pad = k / 2;
    g.transposed = 0; g.zero_far = 0; g.flip = 0;
    g.x_cseg = x->cseg; g.x_segstride = x->seg_stride; g.y_cseg = y->cseg; g.y_segstride = y->seg_stride;
    return g;
}

static bool fwd_in_fused(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype) {
    const ConvGeom g = fwd_geom(x, y, k, stride);
    return mfma_conv_eligible(g.Cin, g.Cout, k, dtype, dtype) && mfma_conv_can_fuse_stats(g);
}

extern "C" size_t ru3d_conv3d_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride,
                                                     int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_fwd_in_workspace_bytes_f16(x, y, k, stride, dtype));
    if (!tensor_ok_split(x) || !tensor_ok(y)) return 0;
    size_t need = ru3d_reduce_workspace_bytes(y);
    if (fwd_in_fused(x, y, k, stride, dtype)) {
        const size_t slab = mfma_conv_stats_slab_bytes(fwd_geom(x, y, k, stride));
        if (slab > need) need = slab;
    }
    const size_t conv_ws = ru3d_conv3d_workspace_bytes(x, y, k, stride, dtype);
    if (conv_ws > need) need = conv_ws;
    return need;
}

extern "C" int ru3d_conv3d_fwd_in(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y,
                                  int k, int stride, int dtype, const float* drop_scale, float* mean, float* scale,
                                  void* ws, size_t ws_bytes, float eps, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_fwd_in_f16(x, w_packed, bias, y, k, stride, dtype, drop_scale, mean, scale, ws, ws_bytes, eps, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok_split(x) && tensor_ok(y) && w_packed && mean && scale && ws, "conv3d_fwd_in: bad argument");
    RU3D_REQUIRE(ws_bytes >= ru3d_conv3d_fwd_in_workspace_bytes(x, y, k, stride, dtype), "conv3d_fwd_in: workspace too small");
    if (fwd_in_fused(x, y, k, stride, dtype)) {
        RU3D_REQUIRE(x->n == y->n && y->d == x->d && y->h == x->h && y->w == x->w, "conv3d_fwd_in: extents mismatch");
        const ConvGeom g = fwd_geom(x, y, k, stride);
        int rc = conv_mfma_launch(x->ptr, w_packed, bias, nullptr, y->ptr, g, as_stream(stream), (float*)ws);
        if (rc) return rc;
        return mfma_conv_stats_finalize(g, (const float*)ws, drop_scale, eps, mean, scale, as_stream(stream));
    }
    // the conv's split-K partials and the statistics pass share the workspace (stream order keeps them apart)
    int rc = ru3d_conv3d_fwd(x, w_packed, bias, nullptr, y, k, stride, dtype, dtype, ws, ws_bytes, stream);
    if (rc) return rc;
    return ru3d_instnorm_stats(y, drop_scale, mean, scale, ws, ws_bytes, eps, dtype, stream);
}

// ---- conv + InstanceNorm statistics + apply (+ residual) + LeakyReLU behind one entry point (reference network.py:
// 405-416: x = lrelu(IN(dropout(conv1(x)))), lrelu(IN(conv2(x)) + skip)).  On the small levels the statistics, their
// finalize and the apply are ONE whole-instance kernel (norm_small.hip) that also sums the conv's split-K slices;
// everywhere else this is ru3d_conv3d_fwd_in followed by ru3d_in_lrelu_fwd.
static bool small_conv_path(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride, int dtype) {
    if (dtype != RU3D_BF16 || k != 3 || stride != 1) return false;
    const ConvGeom g = fwd_geom(x, y, k, stride);
    return mfma_conv_eligible(g.Cin, g.Cout, k, dtype, dtype) && mfma_conv_geometry_ok(g) && !mfma_conv_can_fuse_stats(g);
}

#ifndef RU3D_STORAGE_F16
extern "C" size_t ru3d_conv3d_fwd_in_lrelu_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int k, int stride,
                                                           int dtype) {
    return ru3d_conv3d_fwd_in_workspace_bytes(x, y, k, stride, dtype);
}
#endif

extern "C" int ru3d_conv3d_fwd_in_lrelu(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y,
                                        int k, int stride, int dtype, const float* drop_scale, float* mean, float* scale,
                                        const ru3d_tensor* res, const ru3d_tensor* out, float slope, void* ws,
                                        size_t ws_bytes, float eps, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_fwd_in_lrelu_f16(x, w_packed, bias, y, k, stride, dtype, drop_scale, mean, scale, res, out, slope, ws, ws_bytes, eps, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok_split(x) && tensor_ok(y) && tensor_ok(out) && w_packed && mean && scale && ws,
                 "conv3d_fwd_in_lrelu: bad argument");
    RU3D_REQUIRE(res_ok(res, y) && res_ok(out, y), "conv3d_fwd_in_lrelu: res / out do not have the shape of y");
    const int small = (small_conv_path(x, y, k, stride, dtype) && x->n == y->n && y->d == x->d && y->h == x->h &&
                       y->w == x->w) ? in_small_mode(y, false, res, out) : 0;
    if (small) {
        RU3D_REQUIRE(ws_bytes >= ru3d_conv3d_fwd_in_workspace_bytes(x, y, k, stride, dtype), "conv3d_fwd_in_lrelu: workspace too small");
        const ConvGeom g = fwd_geom(x, y, k, stride);
        int ks = 0;      // the whole-instance kernel sums the conv's split-K slices itself; the two-kernel form reads y
        int rc = conv_mfma_launch(x->ptr, w_packed, bias, nullptr, y->ptr, g, as_stream(stream), nullptr, ws, ws_bytes, nullptr,
                                  0, 0.f, nullptr, 0, nullptr, small == 1 ? &ks : nullptr);
        if (rc) return rc;
        return in_small_fwd_launch(y, ks ? (const float*)ws : nullptr, ks, bias, drop_scale, mean, scale, res, out, ws, eps,
                                   slope, as_stream(stream));
    }
    int rc = ru3d_conv3d_fwd_in(x, w_packed, bias, y, k, stride, dtype, drop_scale, mean, scale, ws, ws_bytes, eps, stream);
    if (rc) return rc;
    return ru3d_in_lrelu_fwd(y, mean, scale, res, out, slope, dtype, stream);
}

// ---- can a decoder ResBlock take cat((up, skip)) as a split tensor through all of its kernels? (see ru3d_tensor)
extern "C" int ru3d_planar_concat_supported(int n, int d, int h, int w, int cseg, int cout, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_planar_concat_supported_f16(n, d, h, w, cseg, cout, dtype));
    static const int mode = getenv("RU3D_PLANAR_CONCAT") ? atoi(getenv("RU3D_PLANAR_CONCAT")) : 1;
    if (!mode || dtype != RU3D_BF16 || n <= 0 || d <= 0 || h <= 0 || w <= 0 || cseg != 32 || cout != 32) return 0;
    const int64_t V = (int64_t)d * h * w;
    // byte offsets of the sliding kernel's staging: plane distance + one sample below 2^31
    if (((int64_t)n * V + V) * cseg * 2 >= (1ll << 31)) return 0;
    SlidePlan sp;
    if (!slide64_conv_plan(n, d, h, w, 2 * cseg, cout, &sp)) return 0;                 // conv1 forward (split input)
    ConvGeom gd;                                                                       // its input-gradient pair (split output)
    gd.N = n; gd.Di = gd.Do = d; gd.Hi = gd.Ho = h; gd.Wi = gd.Wo = w; gd.Cin = cout; gd.Cout = 2 * cseg;
    gd.ldx = cout; gd.ldy = cseg; gd.CoutPad = 2 * cseg; gd.ldr = 0; gd.k = 3; gd.stride = 1; gd.pad = 1;
    gd.transposed = 0; gd.zero_far = 0; gd.flip = 1;
    static const int pair_mode = getenv("RU3D_DGRAD_PAIR") ? atoi(getenv("RU3D_DGRAD_PAIR")) : 1;
    if (!pair_mode || !mfma_conv_eligible(gd.Cin, gd.Cout, 3, dtype, dtype) || !mfma_conv_can_fuse_partner(gd)) return 0;
    WgradGeom gw;                                                                      // conv1's weight gradient (split x)
    gw.N = n; gw.Di = gw.Do = d; gw.Hi = gw.Ho = h; gw.Wi = gw.Wo = w; gw.Cin = 2 * cseg; gw.Cout = cout;
    gw.ldx = cseg; gw.lddy = cout; gw.k = 3; gw.taps = 27; gw.stride = 1; gw.pad = 1;
    gw.s_o = (int64_t)gw.Cin * 27; gw.s_i = 27; gw.chunk_len = 0;
    WgradSlidePlan wp;
    if (!wgrad_slide_plan(gw, &wp)) return 0;
    // the fused tail (skip conv + InstanceNorm apply): skip1x1_fused_eligible's shape conditions
    static const int skip_mode = getenv("RU3D_FUSED_SKIP") ? atoi(getenv("RU3D_FUSED_SKIP")) : 1;
    if (!skip_mode || (V % 16) != 0 || V * n < 65536) return 0;
    return 1;
}

extern "C" int ru3d_conv3d_dgrad(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* res,
                                 const ru3d_tensor* dx, int k, int stride, int dtype, void* ws, size_t ws_bytes,
                                 void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_dgrad_f16(dy, w_packed, res, dx, k, stride, dtype, ws, ws_bytes, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(dy) && tensor_ok(dx) && w_packed, "conv3d_dgrad: bad tensor/weight");
    RU3D_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "conv3d_dgrad: k=%d stride=%d unsupported", k, stride);
    RU3D_REQUIRE(dtype_ok(dtype), "conv3d_dgrad: bad dtype");
    RU3D_REQUIRE(dx->n == dy->n && dy->d == conv_out(dx->d, k, stride) && dy->h == conv_out(dx->h, k, stride) &&
                     dy->w == conv_out(dx->w, k, stride),
                 "conv3d_dgrad: dy extents (%d,%d,%d) do not match dx (%d,%d,%d) k=%d s=%d", dy->d, dy->h, dy->w,
                 dx->d, dx->h, dx->w, k, stride);
    RU3D_REQUIRE(res_ok(res, dx), "conv3d_dgrad: residual shape mismatch");
    // stride 1: gather form with reversed taps; stride 2: transposed (fractionally strided) form
    if (stride == 1)
        return run_conv(dy, w_packed, nullptr, res, dx, k, 1, 0, 1, 0, dtype, dtype, as_stream(stream), ws, ws_bytes);
    return run_conv(dy, w_packed, nullptr, res, dx, k, stride, 1, 0, 0, dtype, dtype, as_stream(stream));
}

// ---- decoder ResBlock: input gradient of conv1 (k3 s1) and skip_conv (k1 s1) in one launch (conv_slide32.hip, 28th tap)
static ConvGeom dgrad_s1_geom(const ru3d_tensor* dy, const ru3d_tensor* da, int k);
extern "C" int ru3d_conv3d_s1_dgrad_pair_supported(const ru3d_tensor* dy, const ru3d_tensor* dy2, const ru3d_tensor* dx, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s1_dgrad_pair_supported_f16(dy, dy2, dx, dtype));
    static const int mode = getenv("RU3D_DGRAD_PAIR") ? atoi(getenv("RU3D_DGRAD_PAIR")) : 1;
    if (!mode || dtype != RU3D_BF16 || !tensor_ok(dy) || !tensor_ok(dy2) || !tensor_ok_split(dx)) return 0;
    if (dx->cseg && (dx->cseg != 32 || dx->c != 64)) return 0;      // split output: two 32-channel slices, one per grid row
    if (dy2->n != dy->n || dy2->d != dy->d || dy2->h != dy->h || dy2->w != dy->w || dy2->c != dy->c) return 0;
    if (dx->n != dy->n || dx->d != dy->d || dx->h != dy->h || dx->w != dy->w) return 0;
    if ((dy->ld % 8) || (dy2->ld % 8) || (dx->ld % 8)) return 0;
    if ((((uintptr_t)dy->ptr) | ((uintptr_t)dy2->ptr) | ((uintptr_t)dx->ptr)) % 16) return 0;
    if ((int64_t)dy->d * dy->h * dy->w * dy->ld >= (1ll << 30)) return 0;
    const ConvGeom g = dgrad_s1_geom(dy, dx, 3);
    return (mfma_conv_eligible(g.Cin, g.Cout, 3, dtype, dtype) && mfma_conv_can_fuse_partner(g)) ? 1 : 0;
}

extern "C" int ru3d_conv3d_s1_dgrad_pair(const ru3d_tensor* dy, const void* w3_packed, const ru3d_tensor* dy2,
                                         const void* w1_packed, const ru3d_tensor* dx, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s1_dgrad_pair_f16(dy, w3_packed, dy2, w1_packed, dx, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(w3_packed && w1_packed, "conv3d_s1_dgrad_pair: null weight");
    RU3D_REQUIRE(ru3d_conv3d_s1_dgrad_pair_supported(dy, dy2, dx, dtype),
                 "conv3d_s1_dgrad_pair: shapes have no fused kernel (ask ru3d_conv3d_s1_dgrad_pair_supported first)");
    const ConvGeom g = dgrad_s1_geom(dy, dx, 3);
    return conv_mfma_launch(dy->ptr, w3_packed, nullptr, nullptr, dx->ptr, g, as_stream(stream), nullptr, nullptr, 0, nullptr, 0,
                            0.f, dy2->ptr, dy2->ld, w1_packed);
}

// ---- decoder ResBlock tail: skip conv + InstanceNorm apply + sum + LeakyReLU (fused_skip.hip)
extern "C" int ru3d_skip1x1_in_lrelu_fwd_supported(const ru3d_tensor* x, const ru3d_tensor* y, const ru3d_tensor* out, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_skip1x1_in_lrelu_fwd_supported_f16(x, y, out, dtype));
    return skip1x1_fused_eligible(x, y, out, dtype) ? 1 : 0;
}

extern "C" int ru3d_skip1x1_in_lrelu_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias, const ru3d_tensor* y,
                                         const float* mean, const float* scale, const ru3d_tensor* out, float slope, int dtype,
                                         void* stream) {
    RU3D_FWD_F16(dtype, ru3d_skip1x1_in_lrelu_fwd_f16(x, w_packed, bias, y, mean, scale, out, slope, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(w_packed && mean && scale, "skip1x1_in_lrelu_fwd: null argument");
    RU3D_REQUIRE(skip1x1_fused_eligible(x, y, out, dtype),
                 "skip1x1_in_lrelu_fwd: shapes have no fused kernel (ask ru3d_skip1x1_in_lrelu_fwd_supported first)");
    return skip1x1_fused_launch(x, w_packed, bias, y, mean, scale, out, slope, as_stream(stream));
}

// ---- the two stride-2 convs of a pooling ResBlock: forward of both + InstanceNorm sums in one launch (conv_s2.hip, G form)
extern "C" int ru3d_conv3d_s2_pair_fwd_in_supported(const ru3d_tensor* x, const ru3d_tensor* y3, const ru3d_tensor* y1, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s2_pair_fwd_in_supported_f16(x, y3, y1, dtype));
    if (dtype != RU3D_BF16 || !tensor_ok(x) || !tensor_ok(y3) || !tensor_ok(y1)) return 0;
    if (y3->n != x->n || y3->d != conv_out(x->d, 3, 2) || y3->h != conv_out(x->h, 3, 2) || y3->w != conv_out(x->w, 3, 2)) return 0;
    if (y1->n != y3->n || y1->d != y3->d || y1->h != y3->h || y1->w != y3->w || y1->c != y3->c) return 0;
    if ((y1->ld % 4) || (int64_t)y1->d * y1->h * y1->w * y1->ld >= (1ll << 30)) return 0;
    if ((((uintptr_t)x->ptr) % 16) || ((((uintptr_t)y3->ptr) | ((uintptr_t)y1->ptr)) % 8)) return 0;
    return conv_s2_tile_eligible(fwd_geom(x, y3, 3, 2)) ? 1 : 0;
}

extern "C" size_t ru3d_conv3d_s2_pair_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y3, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s2_pair_fwd_in_workspace_bytes_f16(x, y3, dtype));
    if (!tensor_ok(x) || !tensor_ok(y3)) return 0;
    return conv_s2_tile_slab_bytes(fwd_geom(x, y3, 3, 2));
}

extern "C" int ru3d_conv3d_s2_pair_fwd_in(const ru3d_tensor* x, const void* w3_packed, const float* b3, const ru3d_tensor* y3,
                                          const void* w1_packed, const float* b1, const ru3d_tensor* y1,
                                          const float* drop_scale, float* mean, float* scale, void* ws, size_t ws_bytes,
                                          float eps, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s2_pair_fwd_in_f16(x, w3_packed, b3, y3, w1_packed, b1, y1, drop_scale, mean, scale, ws, ws_bytes, eps, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(w3_packed && w1_packed && mean && scale && ws, "conv3d_s2_pair_fwd_in: null argument");
    RU3D_REQUIRE(ru3d_conv3d_s2_pair_fwd_in_supported(x, y3, y1, dtype),
                 "conv3d_s2_pair_fwd_in: shapes have no fused kernel (ask ru3d_conv3d_s2_pair_fwd_in_supported first)");
    const ConvGeom g = fwd_geom(x, y3, 3, 2);
    const size_t slab = conv_s2_tile_slab_bytes(g);
    RU3D_REQUIRE(ws_bytes >= slab, "conv3d_s2_pair_fwd_in: workspace too small");
    int rc = conv_s2_tile_launch(x->ptr, w3_packed, b3, y3->ptr, g, (float*)ws, w1_packed, b1, y1->ptr, y1->ld, as_stream(stream));
    if (rc) return rc;
    int gx, cb;
    if (conv_s2_tile_slab_geom(g, &gx, &cb)) return ru3d_fail(-1, "conv3d_s2_pair_fwd_in: no slab geometry");
    return stats_slab_finalize_launch((const float*)ws, gx, cb, g.N, g.Cout, 1.0 / ((double)g.Do * g.Ho * g.Wo), drop_scale, eps,
                                      mean, scale, as_stream(stream));
}

// ---- the two stride-2 convs of a pooling ResBlock: their input gradients in one launch (conv_s2.hip, T form)
static ConvGeom s2_dgrad_geom(const ru3d_tensor* dy, const ru3d_tensor* res, const ru3d_tensor* dx) {
    ConvGeom g;
    g.N = dy->n;
    g.Di = dy->d; g.Hi = dy->h; g.Wi = dy->w; g.Cin = dy->c; g.ldx = dy->ld;
    g.Do = dx->d; g.Ho = dx->h; g.Wo = dx->w; g.Cout = dx->c; g.ldy = dx->ld;
    g.CoutPad = generic_cout_pad(dx->c);
    g.ldr = res ? res->ld : 0;
    g.k = 3; g.stride = 2; g.pad = 1;
    g.transposed = 1; g.zero_far = 0; g.flip = 0;
    return g;
}

extern "C" int ru3d_conv3d_s2_dgrad_pair_supported(const ru3d_tensor* dy, const ru3d_tensor* dy2, const ru3d_tensor* res,
                                                   const ru3d_tensor* dx, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s2_dgrad_pair_supported_f16(dy, dy2, res, dx, dtype));
    if (dtype != RU3D_BF16 || !tensor_ok(dy) || !tensor_ok(dy2) || !tensor_ok(dx) || !res_ok(res, dx)) return 0;
    if (dy2->n != dy->n || dy2->d != dy->d || dy2->h != dy->h || dy2->w != dy->w || dy2->c != dy->c) return 0;
    if (dx->n != dy->n || dy->d != conv_out(dx->d, 3, 2) || dy->h != conv_out(dx->h, 3, 2) || dy->w != conv_out(dx->w, 3, 2))
        return 0;
    if ((dy2->ld % 8) || (int64_t)dy2->d * dy2->h * dy2->w * dy2->ld >= (1ll << 30)) return 0;
    if ((((uintptr_t)dy->ptr) | ((uintptr_t)dy2->ptr) | ((uintptr_t)dx->ptr) | (res ? (uintptr_t)res->ptr : 0)) % 16) return 0;
    return convt_s2_tile_eligible(s2_dgrad_geom(dy, res, dx)) ? 1 : 0;
}

extern "C" int ru3d_conv3d_s2_dgrad_pair(const ru3d_tensor* dy, const void* w3_packed, const ru3d_tensor* dy2,
                                         const void* w1_packed, const ru3d_tensor* res, const ru3d_tensor* dx, int dtype,
                                         void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_s2_dgrad_pair_f16(dy, w3_packed, dy2, w1_packed, res, dx, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(w3_packed && w1_packed, "conv3d_s2_dgrad_pair: null weight");
    RU3D_REQUIRE(ru3d_conv3d_s2_dgrad_pair_supported(dy, dy2, res, dx, dtype),
                 "conv3d_s2_dgrad_pair: shapes have no fused kernel (ask ru3d_conv3d_s2_dgrad_pair_supported first)");
    const ConvGeom g = s2_dgrad_geom(dy, res, dx);
    return convt_s2_tile_launch(dy->ptr, w3_packed, nullptr, res ? res->ptr : nullptr, dx->ptr, g, nullptr, dy2->ptr,
                                dy2->ld, w1_packed, as_stream(stream));
}

// ---- conv input gradient followed by the InstanceNorm + LeakyReLU backward of the tensor it differentiates
static ConvGeom dgrad_s1_geom(const ru3d_tensor* dy, const ru3d_tensor* da, int k) {
    ConvGeom g = fwd_geom(dy, da, k, 1);
    g.flip = 1;
    return g;
}

static bool dgrad_in_bwd_fused(const ru3d_tensor* dy, const ru3d_tensor* da, int k, int stride, int dtype) {
    static const int mode = getenv("RU3D_DGRAD_IN_FUSE") ? atoi(getenv("RU3D_DGRAD_IN_FUSE")) : 1;
    if (!mode || k != 3 || stride != 1 || dtype == RU3D_F32) return false;
    const ConvGeom g = dgrad_s1_geom(dy, da, k);
    return mfma_conv_eligible(g.Cin, g.Cout, k, dtype, dtype) && mfma_conv_can_fuse_bwd_sums(g) &&
           (da->ld % 8) == 0 && (dy->ld % 8) == 0;
}

extern "C" size_t ru3d_conv3d_dgrad_in_bwd_workspace_bytes(const ru3d_tensor* dy, const ru3d_tensor* da, int k,
                                                           int stride, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_dgrad_in_bwd_workspace_bytes_f16(dy, da, k, stride, dtype));
    if (!tensor_ok(dy) || !tensor_ok(da)) return 0;
    size_t need = ru3d_reduce_workspace_bytes(da);
    if (dgrad_in_bwd_fused(dy, da, k, stride, dtype)) {
        // slab of per-(workgroup, wave) sums, then m12[n][c][2]
        const size_t slab = (mfma_conv_stats_slab_bytes(dgrad_s1_geom(dy, da, k)) + 255) / 256 * 256;
        const size_t fused = slab + (size_t)da->n * da->c * 2 * sizeof(float);
        if (fused > need) need = fused;
    }
    if (stride == 1) {
        const size_t conv_ws = ru3d_conv3d_workspace_bytes(dy, da, k, 1, dtype);
        if (conv_ws > need) need = conv_ws;
    }
    return need;
}

extern "C" int ru3d_conv3d_dgrad_in_bwd(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* act,
                                        const float* mean, const float* scale, const ru3d_tensor* da,
                                        const ru3d_tensor* dyn, int k, int stride, float slope, int dtype, void* ws,
                                        size_t ws_bytes, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_dgrad_in_bwd_f16(dy, w_packed, act, mean, scale, da, dyn, k, stride, slope, dtype, ws, ws_bytes, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(dy) && tensor_ok(act) && tensor_ok(da) && tensor_ok(dyn) && w_packed && mean && scale && ws,
                 "conv3d_dgrad_in_bwd: bad argument");
    RU3D_REQUIRE(res_ok(act, da) && res_ok(dyn, da), "conv3d_dgrad_in_bwd: act / dyn do not have the shape of da");
    RU3D_REQUIRE(ws_bytes >= ru3d_conv3d_dgrad_in_bwd_workspace_bytes(dy, da, k, stride, dtype),
                 "conv3d_dgrad_in_bwd: workspace too small");
    if (dgrad_in_bwd_fused(dy, da, k, stride, dtype) && (act->ld % 8) == 0 && (((uintptr_t)act->ptr) % 16) == 0 &&
        (((uintptr_t)da->ptr) % 16) == 0 && (((uintptr_t)dy->ptr) % 16) == 0) {
        RU3D_REQUIRE(da->n == dy->n && da->d == dy->d && da->h == dy->h && da->w == dy->w, "conv3d_dgrad_in_bwd: extents mismatch");
        const ConvGeom g = dgrad_s1_geom(dy, da, k);
        const size_t slab = (mfma_conv_stats_slab_bytes(g) + 255) / 256 * 256;
        float* m12 = (float*)((char*)ws + slab);
        int rc = conv_mfma_launch(dy->ptr, w_packed, nullptr, nullptr, da->ptr, g, as_stream(stream), (float*)ws, nullptr, 0,
                                  act->ptr, act->ld, slope);
        if (rc) return rc;
        rc = mfma_conv_bwd_sums_finalize(g, (const float*)ws, m12, as_stream(stream));
        if (rc) return rc;
        return ru3d_in_lrelu_bwd_apply(da, act, mean, scale, m12, dyn, slope, 0, dtype, stream);
    }
    // the small levels: the whole InstanceNorm backward is one kernel, which also sums the conv's split-K slices (da is
    // then never stored: nobody else reads it)
    const int small = (small_conv_path(dy, da, k, stride, dtype) && da->n == dy->n && da->d == dy->d && da->h == dy->h &&
                       da->w == dy->w) ? in_small_mode(act, false, da, dyn) : 0;
    if (small) {
        const ConvGeom g = dgrad_s1_geom(dy, da, k);
        int ks = 0;
        int rc = conv_mfma_launch(dy->ptr, w_packed, nullptr, nullptr, da->ptr, g, as_stream(stream), nullptr, ws, ws_bytes,
                                  nullptr, 0, 0.f, nullptr, 0, nullptr, small == 1 ? &ks : nullptr);
        if (rc) return rc;
        return in_small_bwd_launch(da, ks ? (const float*)ws : nullptr, ks, act, act, mean, scale, dyn, nullptr, ws, slope, 0,
                                   nullptr, as_stream(stream));
    }
    // not a shape the sliding kernel takes: the two entry points one after the other (they share the workspace in stream order)
    int rc = ru3d_conv3d_dgrad(dy, w_packed, nullptr, da, k, stride, dtype, ws, ws_bytes, stream);
    if (rc) return rc;
    return ru3d_in_lrelu_bwd(da, act, act, mean, scale, dyn, nullptr, ws, ws_bytes, slope, 0, nullptr, nullptr, dtype, stream);
}

static WgradGeom make_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride) {
    WgradGeom g;
    g.N = x->n;
    g.Di = x->d; g.Hi = x->h; g.Wi = x->w; g.Cin = x->c; g.ldx = x->ld;
    g.Do = dy->d; g.Ho = dy->h; g.Wo = dy->w; g.Cout = dy->c; g.lddy = dy->ld;
    g.k = k; g.taps = k * k * k; g.stride = stride; g.pad = k / 2;
    g.s_o = (int64_t)x->c * g.taps;
    g.s_i = g.taps;
    g.chunk_len = 0;
    g.x_cseg = x->cseg; g.x_segstride = x->seg_stride;
    return g;
}

static bool wgrad_shapes_ok(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride) {
    return tensor_ok_split(x) && tensor_ok(dy) && x->n == dy->n && dy->d == conv_out(x->d, k, stride) &&
           dy->h == conv_out(x->h, k, stride) && dy->w == conv_out(x->w, k, stride);
}

extern "C" size_t ru3d_conv3d_wgrad_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride,
                                                    int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_workspace_bytes_f16(x, dy, k, stride, dtype));
    if (!wgrad_shapes_ok(x, dy, k, stride)) return 0;
    WgradGeom g = make_wgrad(x, dy, k, stride);
    if (g.x_cseg) return mfma_wgrad_eligible(g, dtype) ? wgrad_mfma_ws_bytes(g) : 0;
    if (stem_wgrad_eligible(g)) return stem_wgrad_ws_bytes(g);
    if (head_wgrad_eligible(g, dtype)) return head_wgrad_ws_bytes(g);
    if (mfma_wgrad_eligible(g, dtype)) return wgrad_mfma_ws_bytes(g);
    return wgrad_generic_ws_bytes(g);
}

extern "C" int ru3d_conv3d_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, void* ws, size_t ws_bytes,
                                 int k, int stride, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_f16(x, dy, dw, ws, ws_bytes, k, stride, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "conv3d_wgrad: k=%d stride=%d unsupported", k, stride);
    RU3D_REQUIRE(wgrad_shapes_ok(x, dy, k, stride), "conv3d_wgrad: x/dy shape mismatch");
    RU3D_REQUIRE(dw && dtype_ok(dtype), "conv3d_wgrad: bad argument");
    WgradGeom g = make_wgrad(x, dy, k, stride);
    if (g.x_cseg) {      // planar concat: the MFMA kernels only (wgrad_mfma_launch refuses the forms that cannot take it)
        RU3D_REQUIRE(mfma_wgrad_eligible(g, dtype), "conv3d_wgrad: no kernel takes a split x for this shape / dtype");
        return wgrad_mfma_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, as_stream(stream));
    }
    if (stem_wgrad_eligible(g)) return stem_wgrad_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, dtype, as_stream(stream));
    if (head_wgrad_eligible(g, dtype))
        return head_wgrad_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, dtype, as_stream(stream));
    if (mfma_wgrad_eligible(g, dtype)) return wgrad_mfma_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, as_stream(stream));
    return wgrad_generic_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, dtype, as_stream(stream));
}


// ---- a ResBlock's two weight gradients on its input x (conv1: 3x3x3, skip_conv: 1x1x1, both with the block's stride;
// network.py:403-409) from ONE pass over x: the sliding (stride 1: wgrad_slide.hip) and the LDS-DMA (stride 2: wgrad_s2.hip)
// weight-gradient kernels have one tap slot of their four waves' 28 free, and the centre tap's rows are the rows the 1x1x1
// conv reads
static bool wgrad_pair_ok(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, int stride, int dtype) {
    static const int mode = getenv("RU3D_WGRAD_PAIR") ? atoi(getenv("RU3D_WGRAD_PAIR")) : 1;      // 0 = two launches
    if (!mode || dtype != RU3D_BF16 || (stride != 1 && stride != 2)) return false;
    if (!wgrad_shapes_ok(x, dy, 3, stride) || !tensor_ok(dy2)) return false;
    if (dy2->n != dy->n || dy2->d != dy->d || dy2->h != dy->h || dy2->w != dy->w || dy2->c != dy->c) return false;
    if ((dy2->ld % 8) || ((((uintptr_t)x->ptr) | ((uintptr_t)dy->ptr) | ((uintptr_t)dy2->ptr)) % 16)) return false;
    if ((int64_t)dy->d * dy->h * dy->w * dy2->ld >= (1ll << 30)) return false;
    WgradGeom g = make_wgrad(x, dy, 3, stride);
    if (g.x_cseg && (g.x_cseg % 32)) return false;
    if (!mfma_wgrad_eligible(g, dtype)) return false;
    if (stride == 2) return wgrad_s2_pair_eligible(g);
    WgradSlidePlan sp;
    return wgrad_slide_plan(g, &sp);
}

extern "C" int ru3d_conv3d_wgrad_pair_supported(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, int stride,
                                                int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_pair_supported_f16(x, dy, dy2, stride, dtype));
    return (x && dy && dy2 && wgrad_pair_ok(x, dy, dy2, stride, dtype)) ? 1 : 0;
}

extern "C" size_t ru3d_conv3d_wgrad_pair_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2,
                                                         int stride, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_pair_workspace_bytes_f16(x, dy, dy2, stride, dtype));
    if (!x || !dy || !dy2 || !wgrad_pair_ok(x, dy, dy2, stride, dtype)) return 0;
    const WgradGeom g = make_wgrad(x, dy, 3, stride);
    return stride == 2 ? wgrad_s2_pair_ws_bytes(g) : wgrad_slide_pair_ws_bytes(g);
}

extern "C" int ru3d_conv3d_wgrad_pair(const ru3d_tensor* x, const ru3d_tensor* dy, const ru3d_tensor* dy2, float* dw,
                                      float* dw2, void* ws, size_t ws_bytes, int stride, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_pair_f16(x, dy, dy2, dw, dw2, ws, ws_bytes, stride, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(x && dy && dy2 && wgrad_pair_ok(x, dy, dy2, stride, dtype),
                 "conv3d_wgrad_pair: shapes have no fused kernel (ask ru3d_conv3d_wgrad_pair_supported first)");
    RU3D_REQUIRE(dw && dw2 && ws, "conv3d_wgrad_pair: null output / workspace");
    WgradGeom g = make_wgrad(x, dy, 3, stride);
    if (stride == 2) {
        RU3D_REQUIRE(ws_bytes >= wgrad_s2_pair_ws_bytes(g), "conv3d_wgrad_pair: workspace too small");
        return wgrad_s2_pair_launch(x->ptr, dy->ptr, dy2->ptr, dy2->ld, dw, dw2, ws, g, as_stream(stream));
    }
    RU3D_REQUIRE(ws_bytes >= wgrad_slide_pair_ws_bytes(g), "conv3d_wgrad_pair: workspace too small");
    return wgrad_slide_pair_launch(x->ptr, dy->ptr, dy2->ptr, dy2->ld, dw, dw2, ws, g, as_stream(stream));
}

// ---- weight gradient AND bias gradient (= sum of dy over samples and voxels) of a conv behind one entry point: the stem's
// MFMA weight-gradient kernel delivers the bias row from the same pass over dy; elsewhere ru3d_conv3d_wgrad + ru3d_channel_sum
extern "C" size_t ru3d_conv3d_wgrad_bias_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy, int k, int stride,
                                                         int dtype) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_bias_workspace_bytes_f16(x, dy, k, stride, dtype));
    if (!wgrad_shapes_ok(x, dy, k, stride)) return 0;
    const size_t a = ru3d_conv3d_wgrad_workspace_bytes(x, dy, k, stride, dtype), b = ru3d_reduce_workspace_bytes(dy);
    return a > b ? a : b;
}

extern "C" int ru3d_conv3d_wgrad_bias(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, float* db, void* ws,
                                      size_t ws_bytes, int k, int stride, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_conv3d_wgrad_bias_f16(x, dy, dw, db, ws, ws_bytes, k, stride, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE((k == 1 || k == 3) && (stride == 1 || stride == 2), "conv3d_wgrad_bias: k=%d stride=%d unsupported", k, stride);
    RU3D_REQUIRE(wgrad_shapes_ok(x, dy, k, stride) && tensor_ok(x), "conv3d_wgrad_bias: x/dy shape mismatch");
    RU3D_REQUIRE(dw && db && dtype_ok(dtype), "conv3d_wgrad_bias: bad argument");
    RU3D_REQUIRE(ws && ws_bytes >= ru3d_conv3d_wgrad_bias_workspace_bytes(x, dy, k, stride, dtype),
                 "conv3d_wgrad_bias: workspace too small");
    WgradGeom g = make_wgrad(x, dy, k, stride);
    if (stem_wgrad_gives_bias(g, dtype))
        return stem_wgrad_launch(x->ptr, dy->ptr, dw, ws, ws_bytes, g, dtype, as_stream(stream), db);
    int rc = ru3d_conv3d_wgrad(x, dy, dw, ws, ws_bytes, k, stride, dtype, stream);
    if (rc) return rc;
    return ru3d_channel_sum(dy, db, ws, ws_bytes, dtype, stream);
}

// ---- the 1x1x1 head's whole backward (reference network.py:547 fc): dlogits (fp32, as the loss kernel leaves it) -> dx, dW, db
extern "C" int ru3d_head_bwd_supported(const ru3d_tensor* x, const ru3d_tensor* dlogits, const ru3d_tensor* dx, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_head_bwd_supported_f16(x, dlogits, dx, dtype));
    if (!tensor_ok(x) || !tensor_ok(dlogits) || !tensor_ok(dx)) return 0;
    if (x->n != dlogits->n || x->d != dlogits->d || x->h != dlogits->h || x->w != dlogits->w) return 0;
    if (dx->n != x->n || dx->d != x->d || dx->h != x->h || dx->w != x->w || dx->c != x->c) return 0;
    if ((x->ld % 8) || (dx->ld % 8) || ((((uintptr_t)x->ptr) | ((uintptr_t)dx->ptr)) % 16) || (((uintptr_t)dlogits->ptr) % 4)) return 0;
    return head_bwd_eligible(x->c, dlogits->c, dtype) ? 1 : 0;
}

extern "C" size_t ru3d_head_bwd_workspace_bytes(const ru3d_tensor* x, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_head_bwd_workspace_bytes_f16(x, dtype));
    return tensor_ok(x) ? head_bwd_ws_bytes(nvox(x), x->c) : 0;
}

extern "C" int ru3d_head_bwd(const ru3d_tensor* x, const ru3d_tensor* dlogits, const float* weight, int cin_real,
                             const ru3d_tensor* dx, float* dw, float* db, void* ws, size_t ws_bytes, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_head_bwd_f16(x, dlogits, weight, cin_real, dx, dw, db, ws, ws_bytes, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(ru3d_head_bwd_supported(x, dlogits, dx, dtype), "head_bwd: shapes have no fused kernel (ask ru3d_head_bwd_supported first)");
    RU3D_REQUIRE(weight && dw && ws && cin_real > 0 && cin_real <= x->c, "head_bwd: bad argument");
    RU3D_REQUIRE(ws_bytes >= head_bwd_ws_bytes(nvox(x), x->c), "head_bwd: workspace too small");
    return head_bwd_launch(x->ptr, x->ld, (const float*)dlogits->ptr, dlogits->ld, weight, cin_real, x->c, dlogits->c, dx->ptr,
                           dx->ld, dw, db, ws, nvox(x), as_stream(stream));
}

// --------------------------------------------------------------------------- ConvTranspose3d(k3,s2,p1) + far pad
static bool convt_shapes_ok(const ru3d_tensor* x, const ru3d_tensor* y) {
    return tensor_ok(x) && tensor_ok(y) && x->n == y->n && y->d == 2 * x->d && y->h == 2 * x->h && y->w == 2 * x->w;
}

extern "C" int ru3d_convtranspose3d_k3s2p1_fwd(const ru3d_tensor* x, const void* w_packed, const float* bias,
                                               const ru3d_tensor* y, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_fwd_f16(x, w_packed, bias, y, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(convt_shapes_ok(x, y) && w_packed, "convtranspose3d_fwd: y must have extents 2*x");
    RU3D_REQUIRE(dtype_ok(dtype), "convtranspose3d_fwd: bad dtype");
    return run_conv(x, w_packed, bias, nullptr, y, 3, 2, 1, 0, 1, dtype, dtype, as_stream(stream));
}

static ConvGeom convt_fwd_geom(const ru3d_tensor* x, const ru3d_tensor* y) {
    ConvGeom g;
    g.N = x->n;
    g.Di = x->d; g.Hi = x->h; g.Wi = x->w; g.Cin = x->c; g.ldx = x->ld;
    g.Do = y->d; g.Ho = y->h; g.Wo = y->w; g.Cout = y->c; g.ldy = y->ld;
    g.CoutPad = generic_cout_pad(y->c);
    g.ldr = 0;
    g.k = 3; g.stride = 2; g.pad = 1;
    g.transposed = 1; g.zero_far = 1; g.flip = 0;
    return g;
}

static bool convt_fwd_in_fused(const ru3d_tensor* x, const ru3d_tensor* y, int dtype) {
    return dtype == RU3D_BF16 && convt_shapes_ok(x, y) && mfma_conv_eligible(x->c, y->c, 3, dtype, dtype) &&
           ((((uintptr_t)x->ptr) | ((uintptr_t)y->ptr)) % 16) == 0 && convt_s2_tile_eligible(convt_fwd_geom(x, y));
}

extern "C" size_t ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* y, int dtype) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes_f16(x, y, dtype));
    if (!tensor_ok(x) || !tensor_ok(y)) return 0;
    size_t need = ru3d_reduce_workspace_bytes(y);
    if (convt_fwd_in_fused(x, y, dtype)) {
        const size_t slab = convt_s2_tile_slab_bytes(convt_fwd_geom(x, y));
        if (slab > need) need = slab;
    }
    return need;
}

extern "C" int ru3d_convtranspose3d_k3s2p1_fwd_in(const ru3d_tensor* x, const void* w_packed, const float* bias,
                                                  const ru3d_tensor* y, float* mean, float* scale, void* ws,
                                                  size_t ws_bytes, float eps, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_fwd_in_f16(x, w_packed, bias, y, mean, scale, ws, ws_bytes, eps, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(convt_shapes_ok(x, y) && w_packed && mean && scale && ws, "convtranspose3d_fwd_in: bad argument");
    RU3D_REQUIRE(ws_bytes >= ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes(x, y, dtype),
                 "convtranspose3d_fwd_in: workspace too small");
    if (convt_fwd_in_fused(x, y, dtype)) {
        const ConvGeom g = convt_fwd_geom(x, y);
        int rc = convt_s2_tile_launch(x->ptr, w_packed, bias, nullptr, y->ptr, g, (float*)ws, nullptr, 0, nullptr, as_stream(stream));
        if (rc) return rc;
        int gx, cb;
        if (convt_s2_tile_slab_geom(g, &gx, &cb)) return ru3d_fail(-1, "convtranspose3d_fwd_in: no slab geometry");
        return stats_slab_finalize_launch((const float*)ws, gx, cb, g.N, g.Cout, 1.0 / ((double)g.Do * g.Ho * g.Wo), nullptr, eps,
                                          mean, scale, as_stream(stream));
    }
    int rc = ru3d_convtranspose3d_k3s2p1_fwd(x, w_packed, bias, y, dtype, stream);
    if (rc) return rc;
    return ru3d_instnorm_stats(y, nullptr, mean, scale, ws, ws_bytes, eps, dtype, stream);
}

extern "C" int ru3d_convtranspose3d_k3s2p1_dgrad(const ru3d_tensor* dy, const void* w_packed, const ru3d_tensor* dx,
                                                 int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_dgrad_f16(dy, w_packed, dx, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(convt_shapes_ok(dx, dy) && w_packed, "convtranspose3d_dgrad: dy must have extents 2*dx");
    RU3D_REQUIRE(dtype_ok(dtype), "convtranspose3d_dgrad: bad dtype");
    // dx[i] = sum_tap dy[2i - 1 + tap] . W[.,.,tap]: a stride-2 gather conv over dy (far planes of dy are zero)
    return run_conv(dy, w_packed, nullptr, nullptr, dx, 3, 2, 0, 0, 0, dtype, dtype, as_stream(stream));
}

static WgradGeom make_convt_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy) {
    // dW[ci][co][tap] = sum_i x[i][ci] dy[2i-1+tap][co]: wgrad of a stride-2 conv whose gathered
    // operand is dy and whose dense operand is x.
    WgradGeom g = make_wgrad(dy, x, 3, 2);
    g.s_o = (int64_t)dy->c * 27;  // "cout" of this view = x.c = module in_channels (outer dim of W[ci][co][tap])
    g.s_i = 27;
    return g;
}

extern "C" size_t ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes(const ru3d_tensor* x, const ru3d_tensor* dy,
                                                                    int dtype) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes_f16(x, dy, dtype));
    if (!convt_shapes_ok(x, dy)) return 0;
    WgradGeom g = make_convt_wgrad(x, dy);
    if (mfma_wgrad_eligible(g, dtype)) return wgrad_mfma_ws_bytes(g);
    return wgrad_generic_ws_bytes(g);
}

extern "C" int ru3d_convtranspose3d_k3s2p1_wgrad(const ru3d_tensor* x, const ru3d_tensor* dy, float* dw, void* ws,
                                                 size_t ws_bytes, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_convtranspose3d_k3s2p1_wgrad_f16(x, dy, dw, ws, ws_bytes, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(convt_shapes_ok(x, dy), "convtranspose3d_wgrad: dy must have extents 2*x");
    RU3D_REQUIRE(dw && dtype_ok(dtype), "convtranspose3d_wgrad: bad argument");
    WgradGeom g = make_convt_wgrad(x, dy);
    if (mfma_wgrad_eligible(g, dtype)) return wgrad_mfma_launch(dy->ptr, x->ptr, dw, ws, ws_bytes, g, as_stream(stream));
    return wgrad_generic_launch(dy->ptr, x->ptr, dw, ws, ws_bytes, g, dtype, as_stream(stream));
}

}  // namespace RU3D_NS
