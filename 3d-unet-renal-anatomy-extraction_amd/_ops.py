"""Host-side wrappers over the C ABI (one Python function per entry point) and the autograd
Functions that stitch them into the reference's blocks.  No arithmetic happens here: every
tensor value is produced by a HIP kernel of libru3d.so; torch supplies memory and the tape.
"""
import ctypes
import os
import threading
import weakref

import torch

import _native as N
from _native import check, desc, ptr, ref, stream

LRELU_SLOPE = 0.01   # nn.LeakyReLU default negative_slope (reference network.py:386)
IN_EPS = 1e-5        # nn.InstanceNorm3d default eps (reference network.py:384)


# --------------------------------------------------------------------------- thin wrappers
def pack_weight(w, role, dtype, stride=1):
    """w: fp32 parameter in the reference layout (Conv3d [Cout,Cin,k,k,k]; ConvTranspose3d [Cin,Cout,k,k,k])."""
    N.require_device(w, "weight")
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    k = w.shape[2]
    if role in (N.ROLE_CONVT_FWD, N.ROLE_CONVT_DGRAD):
        cin, cout = w.shape[0], w.shape[1]
    else:
        cout, cin = w.shape[0], w.shape[1]
    code = N.dtype_code(dtype)
    if role in (N.ROLE_CONVT_FWD, N.ROLE_CONVT_DGRAD):
        stride = 2
    nbytes = N.lib.ru3d_packed_weight_bytes(cout, cin, k, stride, role, code)
    if nbytes == 0:
        raise N.Ru3dError("ru3d: cannot pack weight of shape %s" % (tuple(w.shape),))
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(N.lib.ru3d_pack_weight(ptr(w), ptr(out), cout, cin, k, stride, role, code, stream()), "pack_weight")
    return out


_PACK_PLANS = {}
_PACK_CACHE = {}
WEIGHTS_EPOCH = [0]   # bumped by optim.Adam.step (its kernel writes the parameters behind torch's back)


def cpad(c):
    """Channel count of the padded activation that carries `c` real channels on the MFMA kernels."""
    return (c + 31) // 32 * 32


def padded_dim(real, seg):
    return (real // seg) * cpad(seg) if seg else real


def seg_of(real, have, nseg=1):
    """Segment size with which a tensor of `have` channels carries a weight dimension of `real` channels as `nseg`
    zero-padded segments (the decoder's concat input is two: 30 | 30 -> 32 | 32); checks that it does."""
    if real % nseg or nseg * cpad(real // nseg) != have:
        raise N.Ru3dError("ru3d: a tensor with %d channels is not the padded form of %d channels in %d segment(s)"
                          % (have, real, nseg))
    return real // nseg


# Training: Unet.forward packs every weight of the pass up front (prepack) - the ~30 per-block launches of a step, most of
# them 12 us of latency around a few hundred KB, become a handful.  spec key -> (pack, weakref(weight), version, epoch,
# stream); the blocks' own pack_weights calls find their packs here.
_PREPACK = {}


def _spec_key(sp, code):
    w, role, stride, cs, ci = sp
    return (code, w.data_ptr(), tuple(w.shape), role, stride, cs, ci)


def _prepacked(specs, code):
    if not _PREPACK:
        return None
    st = None
    outs = []
    for sp in specs:
        hit = _PREPACK.get(_spec_key(sp, code))
        if hit is None:
            return None
        pack, wref, ver, epoch, stream_id = hit
        w = sp[0]
        if st is None:
            st = torch.cuda.current_stream(w.device).cuda_stream
        if wref() is not w or ver != w._version or epoch != WEIGHTS_EPOCH[0] or stream_id != st:
            return None
        outs.append(pack)
    return outs


def prepack(specs, dtype):
    """Pack every (tensor, role, stride, cout_seg, cin_seg) of `specs` now (duplicates dropped; 3x3x3 weights first, so
    that the two roles of a weight land in the same launch of the pair kernel) and keep the packs for the pass."""
    _PREPACK.clear()
    code = N.dtype_code(dtype)
    seen, uniq = set(), []
    for sp in specs:
        sp = sp if len(sp) == 5 else (sp[0], sp[1], sp[2], 0, 0)
        k = _spec_key(sp, code)
        if k not in seen:
            seen.add(k)
            uniq.append(sp)
    if not uniq:
        return

    def order(sp):
        w, role = sp[0], sp[1]
        if role == N.ROLE_BIAS:
            return (2, 0)
        return (0 if w.shape[2] == 3 else 1, 0)
    uniq.sort(key=order)           # stable: both roles of a weight stay adjacent
    packs = pack_weights(uniq, dtype)
    st = torch.cuda.current_stream(uniq[0][0].device).cuda_stream
    for sp, pk in zip(uniq, packs):
        _PREPACK[_spec_key(sp, code)] = (pk, weakref.ref(sp[0]), sp[0]._version, WEIGHTS_EPOCH[0], st)


def pack_weights(specs, dtype):
    """specs: list of (tensor, role, stride[, cout_seg, cin_seg]) -> list of packed tensors, produced by ONE kernel
    launch per RU3D_PACK_MAX items (the packs share one allocation).  role ROLE_BIAS pads a bias vector
    ([cout] -> fp32 [padded cout]).  The size/offset plan and the ctypes item arrays of a spec list are built once
    and reused while the parameters stay where they are (same data pointers)."""
    code = N.dtype_code(dtype)
    specs = [sp if len(sp) == 5 else (sp[0], sp[1], sp[2], 0, 0) for sp in specs]
    pre = _prepacked(specs, code)
    if pre is not None:
        return pre
    key = (code,) + tuple((w.data_ptr(), w.shape, role, stride, cs, ci) for w, role, stride, cs, ci in specs)
    plan = _PACK_PLANS.get(key)
    if plan is None:
        metas, sizes = [], []
        for w, role, stride, cout_seg, cin_seg in specs:
            N.require_device(w, "weight")
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise N.Ru3dError("ru3d: weights must be contiguous float32 parameters")
            if role == N.ROLE_BIAS:
                cout, cin, k, stride = w.shape[0], 1, 1, 1
                nbytes = 4 * padded_dim(cout, cout_seg)
            else:
                k = w.shape[2]
                if role in (N.ROLE_CONVT_FWD, N.ROLE_CONVT_DGRAD):
                    cin, cout, stride = w.shape[0], w.shape[1], 2
                else:
                    cout, cin = w.shape[0], w.shape[1]
                nbytes = N.lib.ru3d_packed_weight_bytes(padded_dim(cout, cout_seg), padded_dim(cin, cin_seg), k, stride,
                                                        role, code)
            if nbytes == 0:
                raise N.Ru3dError("ru3d: cannot pack weight of shape %s" % (tuple(w.shape),))
            metas.append((w.data_ptr(), cout, cin, k, stride, role, cout_seg, cin_seg))
            sizes.append((nbytes + 255) // 256 * 256)
        offs, off = [], 0
        for sz in sizes:
            offs.append(off)
            off += sz
        chunks = []
        for i0 in range(0, len(metas), N.PACK_MAX):
            chunk = metas[i0:i0 + N.PACK_MAX]
            items = (N.PackItem * len(chunk))()
            for j, (src, cout, cin, k, stride, role, cout_seg, cin_seg) in enumerate(chunk):
                items[j] = N.PackItem(src, 0, cout, cin, k, stride, role, cout_seg, cin_seg)
            chunks.append((i0, items))
        plan = (sizes, offs, off, chunks)
        if len(_PACK_PLANS) > 4096:
            _PACK_PLANS.clear()
        _PACK_PLANS[key] = plan
    sizes, offs, total, chunks = plan
    # Unchanged weights (inference: hundreds of windows through the same model) reuse the last packs.  "Unchanged" =
    # torch's version counters (every in-place torch op / load_state_dict bumps them) and WEIGHTS_EPOCH, which the
    # fused optimizer bumps because it updates parameters through raw pointers.
    dev0 = specs[0][0].device
    N.note_device(dev0)
    ver = tuple(sp[0]._version for sp in specs) + (WEIGHTS_EPOCH[0], torch.cuda.current_stream(dev0).cuda_stream)
    use_cache = not torch.is_grad_enabled()      # inference only: training repacks after every optimizer step anyway
    hit = _PACK_CACHE.get(key) if use_cache else None
    # the entry belongs to these very tensor objects (a freed parameter's address and version can both recur)
    if hit is not None and hit[0] == ver and all(r() is sp[0] for r, sp in zip(hit[2], specs)):
        return hit[1]
    buf = torch.empty(total, dtype=torch.uint8, device=dev0)
    base = buf.data_ptr()
    outs = [buf[o:o + sz] for o, sz in zip(offs, sizes)]
    st = stream()
    for i0, items in chunks:
        for j in range(len(items)):
            items[j].dst = base + offs[i0 + j]
        check(N.lib.ru3d_pack_weights(items, len(items), code, st), "pack_weights")
    if use_cache:
        if len(_PACK_CACHE) > 4096:
            _PACK_CACHE.clear()
        _PACK_CACHE[key] = (ver, outs, [weakref.ref(sp[0]) for sp in specs])
    return outs


def unpad_wgrad(dw_p, cout, cin, cout_seg, cin_seg):
    """Weight gradient computed on padded channels ([cout_p, cin_p, k,k,k]) -> the parameter's shape."""
    if not cout_seg and not cin_seg:
        return dw_p
    k = dw_p.shape[2]
    out = torch.empty((cout, cin, k, k, k), dtype=torch.float32, device=dw_p.device)
    N.note_device(dw_p.device)
    check(N.lib.ru3d_unpad_weight_grad(ptr(dw_p), ptr(out), cout, cin, k * k * k, cout_seg, cin_seg, stream()),
          "unpad_weight_grad")
    return out


def _bias(b):
    if b is None:
        return None
    b = b.detach()
    return b if (b.dtype == torch.float32 and b.is_contiguous()) else b.float().contiguous()


def _conv_out(size, k, s):
    return (size + 2 * (k // 2) - k) // s + 1


class Probe:
    """bench.py: times the main kernel of every 3x3x3 stride-1 conv launch of ONE shape - forward and input gradient,
    through whichever entry point - with HIP-event pairs the library records on the launch stream right around that
    kernel (ru3d_probe_begin / _end; not around the memset / finalize / apply launches that share an entry point)."""

    def __init__(self, n, cin, cout, extent):
        self.key = (int(n),) + tuple(int(e) for e in extent) + (int(cin), int(cout))
        self.count = 0
        self.total_ms = 0.0
        self.active = False

    def start(self):
        check(N.lib.ru3d_probe_begin(*self.key), "probe_begin")
        self.active = True

    def stop(self):
        if not self.active:
            return
        n = ctypes.c_int(0)
        ms = ctypes.c_double(0.0)
        check(N.lib.ru3d_probe_end(ctypes.byref(n), ctypes.byref(ms)), "probe_end")
        self.count += n.value
        self.total_ms += ms.value
        self.active = False

    def result(self):
        self.stop()
        return self.count, (self.total_ms / self.count if self.count else 0.0)


def set_probe(p):
    """Arm (a Probe) or disarm (None) the library's kernel probe."""
    cur = _PROBE[0]
    if cur is not None and cur is not p:
        cur.stop()
    _PROBE[0] = p
    if p is not None:
        p.start()


_PROBE = [None]


def _conv_ws(dsrc, ddst, k, stride, dtype, device):
    """(pointer, bytes) of the optional conv scratch (split-K partials on the deepest level), caller-owned."""
    if k != 3 or stride != 1 or dtype == torch.float32:
        return None, 0
    need = N.lib.ru3d_conv3d_workspace_bytes(ref(dsrc), ref(ddst), k, stride, N.dtype_code(dtype))
    if need == 0:
        return None, 0
    buf = N.workspace(need, device)
    return ptr(buf), buf.numel()


def conv_fwd(x, pw, bias, cout, k, stride, res=None, out_dtype=None):
    n, _, d, h, w = x.shape
    out_dtype = out_dtype or x.dtype
    y = N.new_act(n, cout, _conv_out(d, k, stride), _conv_out(h, k, stride), _conv_out(w, k, stride), out_dtype,
                  x.device)
    b = _bias(bias)
    dx, dyy = desc(x), desc(y)
    dr = desc(res) if res is not None else None
    ws, wsn = _conv_ws(dx, dyy, k, stride, x.dtype, x.device)
    check(N.lib.ru3d_conv3d_fwd(ref(dx), ptr(pw), ptr(b), ref(dr), ref(dyy), k, stride, N.dtype_code(x.dtype),
                                N.dtype_code(out_dtype), ws, wsn, stream()), "conv3d_fwd")
    return y


def conv_fwd_in(x, pw, bias, cout, k, stride, drop_scale=None):
    """conv + InstanceNorm statistics of its output in one entry point (fused in the MFMA epilogue when possible)."""
    n, _, d, h, w = x.shape
    y = N.new_act(n, cout, _conv_out(d, k, stride), _conv_out(h, k, stride), _conv_out(w, k, stride), x.dtype, x.device)
    mean = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    scale = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    b = _bias(bias)
    dx, dyy = desc(x), desc(y)
    code = N.dtype_code(x.dtype)
    ws = N.workspace(N.lib.ru3d_conv3d_fwd_in_workspace_bytes(ref(dx), ref(dyy), k, stride, code), x.device)
    check(N.lib.ru3d_conv3d_fwd_in(ref(dx), ptr(pw), ptr(b), ref(dyy), k, stride, code, ptr(drop_scale), ptr(mean),
                                   ptr(scale), ptr(ws), ws.numel(), IN_EPS, stream()), "conv3d_fwd_in")
    return y, mean, scale


def conv_fwd_in_act(x, pw, bias, cout, k, stride, drop_scale=None, res=None, out=None):
    """conv + InstanceNorm statistics + lrelu(IN(y) (+ res)) behind one entry point (ru3d_conv3d_fwd_in_lrelu): returns
    (y, mean, scale, act).  One launch for statistics + finalize + apply on the small levels; elsewhere conv_fwd_in +
    in_lrelu_fwd inside the library."""
    n, _, d, h, w = x.shape
    od, oh, ow = _conv_out(d, k, stride), _conv_out(h, k, stride), _conv_out(w, k, stride)
    y = N.new_act(n, cout, od, oh, ow, x.dtype, x.device)
    if out is None:
        out = N.new_act(n, cout, od, oh, ow, x.dtype, x.device)
    mean = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    scale = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    b = _bias(bias)
    dx, dyy, do = desc(x), desc(y), desc(out)
    dr = desc(res) if res is not None else None
    code = N.dtype_code(x.dtype)
    ws = N.workspace(N.lib.ru3d_conv3d_fwd_in_lrelu_workspace_bytes(ref(dx), ref(dyy), k, stride, code), x.device)
    check(N.lib.ru3d_conv3d_fwd_in_lrelu(ref(dx), ptr(pw), ptr(b), ref(dyy), k, stride, code, ptr(drop_scale), ptr(mean),
                                         ptr(scale), ref(dr), ref(do), LRELU_SLOPE, ptr(ws), ws.numel(), IN_EPS, stream()),
          "conv3d_fwd_in_lrelu")
    return y, mean, scale, out


def conv_dgrad(dy, pw, in_shape, k, stride, res=None):
    n, cin, d, h, w = in_shape
    dx = N.new_act(n, cin, d, h, w, dy.dtype, dy.device)
    ddy, ddx = desc(dy), desc(dx)
    dr = desc(res) if res is not None else None
    ws, wsn = _conv_ws(ddy, ddx, k, stride, dy.dtype, dy.device)
    check(N.lib.ru3d_conv3d_dgrad(ref(ddy), ptr(pw), ref(dr), ref(ddx), k, stride, N.dtype_code(dy.dtype), ws, wsn,
                                  stream()), "conv3d_dgrad")
    return dx


def conv_dgrad_in_bwd(dy, pw, act, mean, scale, k=3, stride=1):
    """da = conv^T(dy) followed by the InstanceNorm + LeakyReLU backward of act = lrelu(IN(y)): returns d/dy.  On the
    sliding-kernel shapes the backward's two sums come out of the conv's epilogue (ru3d_conv3d_dgrad_in_bwd)."""
    n, c, d, h, w = act.shape
    da = N.new_act(n, c, d, h, w, dy.dtype, dy.device)
    dyn = N.new_act(n, c, d, h, w, dy.dtype, dy.device)
    ddy, dact, dda, ddyn = desc(dy), desc(act), desc(da), desc(dyn)
    code = N.dtype_code(dy.dtype)
    need = N.lib.ru3d_conv3d_dgrad_in_bwd_workspace_bytes(ref(ddy), ref(dda), k, stride, code)
    ws = N.workspace(need, dy.device)
    check(N.lib.ru3d_conv3d_dgrad_in_bwd(ref(ddy), ptr(pw), ref(dact), ptr(mean), ptr(scale), ref(dda), ref(ddyn), k,
                                         stride, LRELU_SLOPE, code, ptr(ws), ws.numel(), stream()),
          "conv3d_dgrad_in_bwd")
    return dyn


# parallel.GradSync: parameter storage address -> the fp32 slice of the exchange bucket its gradient will be sent from.
# The weight-gradient kernels write there directly, so no copy into the bucket is needed (387 MB per step at config 2).
GRAD_ARENA = {}


def _grad_out(key, shape, device):
    t = GRAD_ARENA.get(key) if key is not None else None
    if t is not None and tuple(t.shape) == tuple(shape) and t.device == device:
        return t.view(t.shape)      # a tensor object of its own: autograd adopts a gradient only when nobody else holds it
    return torch.empty(shape, dtype=torch.float32, device=device)


def conv_wgrad(x, dy, k, stride, key=None):
    cout, cin = dy.shape[1], x.shape[1]
    dw = _grad_out(key, (cout, cin, k, k, k), x.device)
    dx, ddy = desc(x), desc(dy)
    code = N.dtype_code(x.dtype)
    nbytes = N.lib.ru3d_conv3d_wgrad_workspace_bytes(ref(dx), ref(ddy), k, stride, code)
    ws = N.workspace(nbytes, x.device)
    check(N.lib.ru3d_conv3d_wgrad(ref(dx), ref(ddy), ptr(dw), ptr(ws), ws.numel(), k, stride, code, stream()),
          "conv3d_wgrad")
    return dw


def _wgrad_pair_ok(x, dy, stride=1):
    """ru3d_conv3d_wgrad_pair takes (x, dy, dy2) with dy2 shaped like dy: asked with dy in both places"""
    if x.dtype == torch.float32:
        return False
    dx, ddy = desc(x), desc(dy)
    return bool(N.lib.ru3d_conv3d_wgrad_pair_supported(ref(dx), ref(ddy), ref(ddy), stride, N.dtype_code(x.dtype)))


def conv_wgrad_pair(x, dy, dy2, stride=1, key=None, key2=None):
    """(dW of the 3x3x3 conv on x with gradient dy, dW of a 1x1x1 conv of the same stride on the same x with gradient dy2)
    from one pass over x (ru3d_conv3d_wgrad_pair: a ResBlock's conv1 + skip_conv), or None when the shapes have no fused
    kernel."""
    if x.dtype == torch.float32 or tuple(dy.shape) != tuple(dy2.shape):
        return None
    dx, ddy, ddy2 = desc(x), desc(dy), desc(dy2)
    code = N.dtype_code(x.dtype)
    if not N.lib.ru3d_conv3d_wgrad_pair_supported(ref(dx), ref(ddy), ref(ddy2), stride, code):
        return None
    cout, cin = dy.shape[1], x.shape[1]
    dw = _grad_out(key, (cout, cin, 3, 3, 3), x.device)
    dw2 = _grad_out(key2, (cout, cin, 1, 1, 1), x.device)
    ws = N.workspace(N.lib.ru3d_conv3d_wgrad_pair_workspace_bytes(ref(dx), ref(ddy), ref(ddy2), stride, code), x.device)
    check(N.lib.ru3d_conv3d_wgrad_pair(ref(dx), ref(ddy), ref(ddy2), ptr(dw), ptr(dw2), ptr(ws), ws.numel(), stride, code,
                                       stream()), "conv3d_wgrad_pair")
    return dw, dw2


def conv_wgrad_bias(x, dy, k, stride, key=None):
    """(dW, db) of a conv with bias from one entry point (the stem's kernel sums dy while it reads it)."""
    cout, cin = dy.shape[1], x.shape[1]
    dw = _grad_out(key, (cout, cin, k, k, k), x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device)
    dx, ddy = desc(x), desc(dy)
    code = N.dtype_code(x.dtype)
    ws = N.workspace(N.lib.ru3d_conv3d_wgrad_bias_workspace_bytes(ref(dx), ref(ddy), k, stride, code), x.device)
    check(N.lib.ru3d_conv3d_wgrad_bias(ref(dx), ref(ddy), ptr(dw), ptr(db), ptr(ws), ws.numel(), k, stride, code, stream()),
          "conv3d_wgrad_bias")
    return dw, db


def head_bwd(x, gy, weight, want_bias, key=None):
    """The 1x1x1 head's backward in one pass (ru3d_head_bwd): gy = dlogits as the loss left them (fp32, NDHWC) ->
    (dx, dW, db), or None when the shapes have no fused kernel."""
    if gy.dtype != torch.float32 or x.dtype == torch.float32 or not N.is_ndhwc(gy) or weight.dtype != torch.float32:
        return None
    n, c, d, h, w = x.shape
    dx = N.new_act(n, c, d, h, w, x.dtype, x.device)
    ddx, dg, dxx = desc(x), desc(gy), desc(dx)
    code = N.dtype_code(x.dtype)
    if not N.lib.ru3d_head_bwd_supported(ref(ddx), ref(dg), ref(dxx), code):
        return None
    cout, cin = weight.shape[0], weight.shape[1]
    dw = _grad_out(key, (cout, cin, 1, 1, 1), x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    ws = N.workspace(N.lib.ru3d_head_bwd_workspace_bytes(ref(ddx), code), x.device)
    wd = weight.detach()
    check(N.lib.ru3d_head_bwd(ref(ddx), ref(dg), ptr(wd if wd.is_contiguous() else wd.contiguous()), cin, ref(dxx), ptr(dw),
                              ptr(db), ptr(ws), ws.numel(), code, stream()), "head_bwd")
    return dx, dw, db


def convt_fwd(x, pw, bias, cout):
    n, _, d, h, w = x.shape
    y = N.new_act(n, cout, 2 * d, 2 * h, 2 * w, x.dtype, x.device)
    b = _bias(bias)
    dx, dyy = desc(x), desc(y)
    check(N.lib.ru3d_convtranspose3d_k3s2p1_fwd(ref(dx), ptr(pw), ptr(b), ref(dyy), N.dtype_code(x.dtype), stream()),
          "convtranspose3d_fwd")
    return y


def convt_fwd_in(x, pw, bias, cout):
    """ConvTranspose3d + far pad and the InstanceNorm statistics of its output in one entry point (fused in the
    transposed conv's epilogue where the stride-2 tile kernel takes the shape)."""
    n, _, d, h, w = x.shape
    y = N.new_act(n, cout, 2 * d, 2 * h, 2 * w, x.dtype, x.device)
    mean = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    scale = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    b = _bias(bias)
    dx, dyy = desc(x), desc(y)
    code = N.dtype_code(x.dtype)
    ws = N.workspace(N.lib.ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes(ref(dx), ref(dyy), code), x.device)
    check(N.lib.ru3d_convtranspose3d_k3s2p1_fwd_in(ref(dx), ptr(pw), ptr(b), ref(dyy), ptr(mean), ptr(scale), ptr(ws),
                                                   ws.numel(), IN_EPS, code, stream()), "convtranspose3d_fwd_in")
    return y, mean, scale


def conv_s2_pair_fwd_in(x, pw3, b3, pw1, b1, cout, drop_scale=None):
    """conv1 (k3 s2) with its InstanceNorm statistics and skip_conv (k1 s2) of the same input in one launch:
    (y3, mean, scale, y1), or None when the shapes have no fused kernel."""
    code = N.dtype_code(x.dtype)
    if code == N.F32:
        return None
    n, _, d, h, w = x.shape
    od, oh, ow = _conv_out(d, 3, 2), _conv_out(h, 3, 2), _conv_out(w, 3, 2)
    y3 = N.new_act(n, cout, od, oh, ow, x.dtype, x.device)
    y1 = N.new_act(n, cout, od, oh, ow, x.dtype, x.device)
    dx, d3, d1 = desc(x), desc(y3), desc(y1)
    if not N.lib.ru3d_conv3d_s2_pair_fwd_in_supported(ref(dx), ref(d3), ref(d1), code):
        return None
    mean = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    scale = torch.empty(n * cout, dtype=torch.float32, device=x.device)
    ws = N.workspace(N.lib.ru3d_conv3d_s2_pair_fwd_in_workspace_bytes(ref(dx), ref(d3), code), x.device)
    check(N.lib.ru3d_conv3d_s2_pair_fwd_in(ref(dx), ptr(pw3), ptr(_bias(b3)), ref(d3), ptr(pw1), ptr(_bias(b1)), ref(d1),
                                           ptr(drop_scale), ptr(mean), ptr(scale), ptr(ws), ws.numel(), IN_EPS, code,
                                           stream()), "conv3d_s2_pair_fwd_in")
    return y3, mean, scale, y1


def conv_s1_dgrad_pair(dy, pw3, dy2, pw1, in_shape, planar=False):
    """Input gradient of a decoder ResBlock's conv1 (k3 s1) and skip_conv (k1 s1) in one launch; None when the shapes
    have no fused kernel.  planar: the gradient of a split concat input - returned as the [2N, C/2, D, H, W] tensor of
    its two planes (N.Split)."""
    n, cin, d, h, w = in_shape
    code = N.dtype_code(dy.dtype)
    if code == N.F32:
        return None
    dx = N.new_act(2 * n, cin // 2, d, h, w, dy.dtype, dy.device) if planar else N.new_act(n, cin, d, h, w, dy.dtype, dy.device)
    ddy, ddy2, ddx = desc(dy), desc(dy2), desc(N.Split(dx) if planar else dx)
    if not N.lib.ru3d_conv3d_s1_dgrad_pair_supported(ref(ddy), ref(ddy2), ref(ddx), code):
        return None
    check(N.lib.ru3d_conv3d_s1_dgrad_pair(ref(ddy), ptr(pw3), ref(ddy2), ptr(pw1), ref(ddx), code, stream()),
          "conv3d_s1_dgrad_pair")
    return dx


def conv_s2_dgrad_pair(dy, pw3, dy2, pw1, in_shape, res=None):
    """Input gradient of a pooling ResBlock's conv1 (k3 s2) and skip_conv (k1 s2) plus `res` in one launch; None when
    the shapes have no fused kernel (the caller then chains two conv_dgrad calls)."""
    n, cin, d, h, w = in_shape
    code = N.dtype_code(dy.dtype)
    if code == N.F32:
        return None
    dx = N.new_act(n, cin, d, h, w, dy.dtype, dy.device)
    ddy, ddy2, ddx = desc(dy), desc(dy2), desc(dx)
    dr = desc(res) if res is not None else None
    if not N.lib.ru3d_conv3d_s2_dgrad_pair_supported(ref(ddy), ref(ddy2), ref(dr), ref(ddx), code):
        return None
    check(N.lib.ru3d_conv3d_s2_dgrad_pair(ref(ddy), ptr(pw3), ref(ddy2), ptr(pw1), ref(dr), ref(ddx), code, stream()),
          "conv3d_s2_dgrad_pair")
    return dx


def convt_dgrad(dy, pw, in_shape):
    n, cin, d, h, w = in_shape
    dx = N.new_act(n, cin, d, h, w, dy.dtype, dy.device)
    ddy, ddx = desc(dy), desc(dx)
    check(N.lib.ru3d_convtranspose3d_k3s2p1_dgrad(ref(ddy), ptr(pw), ref(ddx), N.dtype_code(dy.dtype), stream()),
          "convtranspose3d_dgrad")
    return dx


def convt_wgrad(x, dy, key=None):
    cin, cout = x.shape[1], dy.shape[1]
    dw = _grad_out(key, (cin, cout, 3, 3, 3), x.device)
    dx, ddy = desc(x), desc(dy)
    code = N.dtype_code(x.dtype)
    nbytes = N.lib.ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes(ref(dx), ref(ddy), code)
    ws = N.workspace(nbytes, x.device)
    check(N.lib.ru3d_convtranspose3d_k3s2p1_wgrad(ref(dx), ref(ddy), ptr(dw), ptr(ws), ws.numel(), code, stream()),
          "convtranspose3d_wgrad")
    return dw


def in_stats(y, drop_scale=None):
    n, c = y.shape[0], y.shape[1]
    mean = torch.empty(n * c, dtype=torch.float32, device=y.device)
    scale = torch.empty(n * c, dtype=torch.float32, device=y.device)
    dy = desc(y)
    ws = N.workspace(N.lib.ru3d_reduce_workspace_bytes(ref(dy)), y.device)
    check(N.lib.ru3d_instnorm_stats(ref(dy), ptr(drop_scale), ptr(mean), ptr(scale), ptr(ws), ws.numel(), IN_EPS,
                                    N.dtype_code(y.dtype), stream()), "instnorm_stats")
    return mean, scale


def in_lrelu_fwd(y, mean, scale, res=None, out=None):
    if out is None:
        n, c, d, h, w = y.shape
        out = N.new_act(n, c, d, h, w, y.dtype, y.device)
    dy, do = desc(y), desc(out)
    dr = desc(res) if res is not None else None
    check(N.lib.ru3d_in_lrelu_fwd(ref(dy), ptr(mean), ptr(scale), ref(dr), ref(do), LRELU_SLOPE,
                                  N.dtype_code(y.dtype), stream()), "in_lrelu_fwd")
    return out


def skip1x1_in_lrelu_fwd(x, pw, bias, y, mean, scale, out=None):
    """lrelu(IN(y) + conv1x1(x)) without storing the skip conv's output; None when the shapes have no fused kernel."""
    code = N.dtype_code(y.dtype)
    if code == N.F32:
        return None
    if out is None:
        n, c, d, h, w = y.shape
        out = N.new_act(n, c, d, h, w, y.dtype, y.device)
    dx, dy, do = desc(x), desc(y), desc(out)
    if not N.lib.ru3d_skip1x1_in_lrelu_fwd_supported(ref(dx), ref(dy), ref(do), code):
        return None
    check(N.lib.ru3d_skip1x1_in_lrelu_fwd(ref(dx), ptr(pw), ptr(_bias(bias)), ref(dy), ptr(mean), ptr(scale), ref(do),
                                          LRELU_SLOPE, code, stream()), "skip1x1_in_lrelu_fwd")
    return out


def in_lrelu_bwd(gout, out, y, mean, scale, want_gpre=False, zero_far=False, want_gpre_sum=False, dy_sum=None):
    """dy_sum: optional float32 [C] tensor that receives the channel sums of dy (the producing conv's bias gradient)."""
    n, c, d, h, w = y.shape
    dy = N.new_act(n, c, d, h, w, y.dtype, y.device)
    gpre = N.new_act(n, c, d, h, w, y.dtype, y.device) if want_gpre else None
    dg, do, dyy, ddy = desc(gout), desc(out), desc(y), desc(dy)
    dp = desc(gpre) if want_gpre else None
    ws = N.workspace(N.lib.ru3d_reduce_workspace_bytes(ref(dyy)), y.device)
    gsum = torch.empty(c, dtype=torch.float32, device=y.device) if (want_gpre and want_gpre_sum) else None
    check(N.lib.ru3d_in_lrelu_bwd(ref(dg), ref(do), ref(dyy), ptr(mean), ptr(scale), ref(ddy), ref(dp), ptr(ws),
                                  ws.numel(), LRELU_SLOPE, 1 if zero_far else 0, ptr(gsum), ptr(dy_sum),
                                  N.dtype_code(y.dtype), stream()), "in_lrelu_bwd")
    if want_gpre_sum:
        return dy, gpre, gsum
    return dy, gpre


# --------------------------------------------------------------------------- BatchNorm3d, training mode
_BN_SYNC = [None, 1]     # (process group or None, world size): set_bn_sync


def set_bn_sync(group=None, enable=True):
    """SyncBN: pool the BatchNorm sums of every rank of `group` (None = the default group) in both directions.
    Off by default - the reference's nn.DataParallel normalises each replica's sub-batch on its own (trainer.py:531-535)."""
    import torch.distributed as dist
    if enable and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _BN_SYNC[0], _BN_SYNC[1] = (group if group is not None else dist.group.WORLD), dist.get_world_size(group)
    else:
        _BN_SYNC[0], _BN_SYNC[1] = None, 1


def _bn_allreduce(pooled):
    if _BN_SYNC[0] is not None:
        import torch.distributed as dist
        dist.all_reduce(pooled, group=_BN_SYNC[0])
    return pooled


def bn_train_stats(y, drop_scale, norm, c_real):
    """Batch statistics of `y` (with Dropout3d's factors folded in), the running averages of `norm` moved (reference
    nn.BatchNorm3d defaults, network.py:38-69) -> (fscale, fshift, a, b, count): out = lrelu(y * fscale + fshift),
    x_hat = y * a + b.  c_real: the module's channel count (y may carry zero pad lanes beyond it)."""
    n, c = y.shape[0], y.shape[1]
    dev = y.device
    dy = desc(y)
    pooled = torch.empty(2 * c, dtype=torch.float64, device=dev)
    ws = N.workspace(N.lib.ru3d_reduce_workspace_bytes(ref(dy)), dev)
    check(N.lib.ru3d_batchnorm_stats_pool(ref(dy), ptr(drop_scale), ptr(pooled), ptr(ws), ws.numel(),
                                          N.dtype_code(y.dtype), stream()), "batchnorm_stats_pool")
    _bn_allreduce(pooled)
    count = float(n * y.shape[2] * y.shape[3] * y.shape[4]) * _BN_SYNC[1]
    track = norm.track_running_stats and norm.running_mean is not None
    momentum = 0.0
    if track:
        norm.num_batches_tracked.add_(1)
        momentum = (1.0 / float(norm.num_batches_tracked)) if norm.momentum is None else float(norm.momentum)
    out = torch.empty(4, n * c, dtype=torch.float32, device=dev)
    gamma = norm.weight.detach() if norm.weight is not None else None
    beta = norm.bias.detach() if norm.bias is not None else None
    for t in (gamma, beta, norm.running_mean if track else None, norm.running_var if track else None):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev):
            raise N.Ru3dError("BatchNorm3d: parameters / running statistics must be contiguous fp32 on %s" % dev)
    check(N.lib.ru3d_batchnorm_stats_finalize(ptr(pooled), n, c, c_real, count, ptr(drop_scale), ptr(gamma), ptr(beta),
                                              float(norm.eps), momentum, ptr(norm.running_mean if track else None),
                                              ptr(norm.running_var if track else None), ptr(out[0]), ptr(out[1]),
                                              ptr(out[2]), ptr(out[3]), stream()), "batchnorm_stats_finalize")
    return out[0], out[1], out[2], out[3], count


def affine_lrelu_fwd(y, scale, shift, res=None, out=None):
    if out is None:
        n, c, d, h, w = y.shape
        out = N.new_act(n, c, d, h, w, y.dtype, y.device)
    dy, do = desc(y), desc(out)
    dr = desc(res) if res is not None else None
    check(N.lib.ru3d_affine_lrelu_fwd(ref(dy), ptr(scale), ptr(shift), ref(dr), ref(do), LRELU_SLOPE,
                                      N.dtype_code(y.dtype), stream()), "affine_lrelu_fwd")
    return out


def bn_lrelu_bwd(gout, out, y, a, b, fscale, count, zero_far=False):
    """Backward of out = lrelu(BN(y) (+ res)): -> dy, gpre (= dL/dres), dgamma, dbeta (this rank's sums, padded width)."""
    n, c, d, h, w = y.shape
    dev = y.device
    dy = N.new_act(n, c, d, h, w, y.dtype, dev)
    gpre = N.new_act(n, c, d, h, w, y.dtype, dev)
    dg, do, dyy, ddy, dp = desc(gout), desc(out), desc(y), desc(dy), desc(gpre)
    pooled = torch.empty(2 * c, dtype=torch.float64, device=dev)
    ws = N.workspace(N.lib.ru3d_reduce_workspace_bytes(ref(dyy)), dev)
    code = N.dtype_code(y.dtype)
    check(N.lib.ru3d_batchnorm_bwd_pool(ref(dg), ref(do), ref(dyy), ptr(a), ptr(b), ref(dp), ptr(pooled), ptr(ws),
                                        ws.numel(), LRELU_SLOPE, code, stream()), "batchnorm_bwd_pool")
    local = pooled.view(c, 2).to(torch.float32)
    dbeta, dgamma = local[:, 0].contiguous(), local[:, 1].contiguous()
    if _BN_SYNC[0] is not None:
        pooled = _bn_allreduce(pooled.clone())
    check(N.lib.ru3d_batchnorm_bwd_apply(ref(dp), ref(dyy), ptr(a), ptr(b), ptr(fscale), ptr(pooled), float(count),
                                         ref(ddy), ptr(ws), ws.numel(), 1 if zero_far else 0, code, stream()),
          "batchnorm_bwd_apply")
    return dy, gpre, dgamma, dbeta


def channel_sum(t):
    out = torch.empty(t.shape[1], dtype=torch.float32, device=t.device)
    dt = desc(t)
    ws = N.workspace(N.lib.ru3d_reduce_workspace_bytes(ref(dt)), t.device)
    check(N.lib.ru3d_channel_sum(ref(dt), ptr(out), ptr(ws), ws.numel(), N.dtype_code(t.dtype), stream()),
          "channel_sum")
    return out


def pointwise(op, a, b=None, c=None, n_out=1):
    """ru3d_pointwise (the attention gate's elementwise pieces): returns o1 or (o1, o2)."""
    n, ch, d, h, w = a.shape
    o1 = N.new_act(n, ch, d, h, w, a.dtype, a.device)
    o2 = N.new_act(n, ch, d, h, w, a.dtype, a.device) if n_out == 2 else None
    da, d1 = desc(a), desc(o1)
    db = desc(b) if b is not None else None
    dc = desc(c) if c is not None else None
    d2 = desc(o2) if o2 is not None else None
    check(N.lib.ru3d_pointwise(op, ref(da), ref(db), ref(dc), ref(d1), ref(d2), LRELU_SLOPE, N.dtype_code(a.dtype),
                               stream()), "pointwise")
    return (o1, o2) if n_out == 2 else o1


def copy_channels(src, dst):
    ds, dd = desc(src), desc(dst)
    check(N.lib.ru3d_copy_channels(ref(ds), ref(dd), N.dtype_code(src.dtype), stream()), "copy_channels")


def concat_channels(a, b):
    """cat((a, b), dim=1) as two channel-slice copies (inference paths without a tape)."""
    n, ca, d, h, w = a.shape
    out = N.new_act(n, ca + b.shape[1], d, h, w, a.dtype, a.device)
    copy_channels(a, out[:, :ca])
    copy_channels(as_grad(b, a.dtype), out[:, ca:])
    return out


def add(a, b):
    n, c, d, h, w = a.shape
    out = N.new_act(n, c, d, h, w, a.dtype, a.device)
    da, db, do = desc(a), desc(b), desc(out)
    check(N.lib.ru3d_add(ref(da), ref(db), ref(do), N.dtype_code(a.dtype), stream()), "add")
    return out


def cast_f32(src, dtype):
    """fp32 NDHWC tensor -> storage dtype (no-op for fp32)."""
    if dtype == torch.float32:
        return src
    n, c, d, h, w = src.shape
    out = N.new_act(n, c, d, h, w, dtype, src.device)
    ds, do = desc(src), desc(out)
    check(N.lib.ru3d_cast_f32(ref(ds), ref(do), N.dtype_code(dtype), stream()), "cast_f32")
    return out


def ncdhw_to_ndhwc(x, dtype):
    """fp32 NCDHW-contiguous [N,C,D,H,W] -> NDHWC tensor of `dtype` (the only layout change on the path)."""
    N.require_device(x, "input")
    x = x.detach()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    n, c, d, h, w = x.shape
    out = N.new_act(n, c, d, h, w, dtype, x.device)
    do = desc(out)
    check(N.lib.ru3d_ncdhw_to_ndhwc(ptr(x), ref(do), N.dtype_code(dtype), stream()), "ncdhw_to_ndhwc")
    return out


def ndhwc_to_ncdhw(x):
    n, c, d, h, w = x.shape
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=x.device)
    dx = desc(x)
    check(N.lib.ru3d_ndhwc_to_ncdhw(ref(dx), ptr(out), N.dtype_code(x.dtype), stream()), "ndhwc_to_ncdhw")
    return out


def _dist_rank():
    d = torch.distributed
    return d.get_rank() if (d.is_available() and d.is_initialized()) else 0


_drop_lock = threading.Lock()
_drop_counter = [0]
# graph.GraphedTrainStep sets this while it captures a step: an int64 device scalar the captured dropout launches add
# to their (frozen) counter offsets, so that replay k draws the masks of eager step k
DROP_OFFSET_BASE = [None]


def dropout_scale(n, c, p, device):
    """Per-(n,c) Dropout3d factor (0 or 1/(1-p)) from the on-device counter-based generator."""
    with _drop_lock:
        offset = _drop_counter[0]
        _drop_counter[0] += n * c
    out = torch.empty(n * c, dtype=torch.float32, device=device)
    N.note_device(out.device)
    # one process per GPU: every rank draws its own masks even when all ranks were seeded alike
    seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * _dist_rank()) & 0xFFFFFFFFFFFFFFFF
    base = DROP_OFFSET_BASE[0]
    if base is not None:
        check(N.lib.ru3d_dropout3d_scale_dev(ptr(out), n * c, float(p), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                                             ptr(base), stream()), "dropout3d_scale_dev")
    else:
        check(N.lib.ru3d_dropout3d_scale(ptr(out), n * c, float(p), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                                         stream()), "dropout3d_scale")
    return out


# One launch for all the Dropout3d draws of a forward pass: the generator is counter based (value i of the process is a
# function of (seed, i) alone), so the blocks' factors are consecutive slices of one draw - the same numbers as one
# launch per block, 18 launches fewer per step at config 2.
_DROP_POOL = []


def prefill_dropout(specs, device):
    """specs: [(n, c, p)] of the ResBlocks that will draw, in execution order (all with the same p)."""
    _DROP_POOL.clear()
    if len(specs) < 2 or any(sp[2] != specs[0][2] for sp in specs):
        return
    flat = dropout_scale(1, sum(n * c for n, c, _ in specs), specs[0][2], device)
    off = 0
    for n, c, p in specs:
        _DROP_POOL.append((flat[off:off + n * c], n, c, p))
        off += n * c


def take_dropout(n, c, p):
    if not _DROP_POOL:
        return None
    t, n0, c0, p0 = _DROP_POOL[0]
    if (n0, c0, p0) != (n, c, p):      # not the pass the pool was filled for
        _DROP_POOL.clear()
        return None
    _DROP_POOL.pop(0)
    return t


def as_input(x, dtype):
    """Bring a user tensor onto the native path: NDHWC memory in the storage dtype.  C == 1 inputs in the
    reference's NCDHW layout are already NDHWC; anything else goes through the repack kernel."""
    N.require_device(x, "input")
    if x.dim() != 5:
        raise N.Ru3dError("ru3d: expected a 5-D [N,C,D,H,W] tensor, got %s" % (tuple(x.shape),))
    if x.dtype == dtype and N.is_ndhwc(x):
        return x
    if N.is_ndhwc(x) and x.dtype == torch.float32:
        return cast_f32(x, dtype)
    if x.is_contiguous() or x.dtype != torch.float32:
        return ncdhw_to_ndhwc(x, dtype)
    return ncdhw_to_ndhwc(x.contiguous(), dtype)


def as_grad(g, like_dtype):
    """Gradient tensors arriving from autograd: make them NDHWC in the storage dtype (plumbing only)."""
    if g.dtype != like_dtype:
        if g.dtype == torch.float32 and N.is_ndhwc(g):
            return cast_f32(g, like_dtype)
        g = g.to(like_dtype)
    return N.to_ndhwc(g)


# --------------------------------------------------------------------------- second stream for weight gradients
# Inside one block's backward the weight gradient of a conv and its input gradient are independent: on the SMALL levels
# (16^3 / 8^3 voxels: kernels of 20-50 us that fill a fraction of the chip) the wgrad launches go to a
# second HIP stream and run beside the dgrad -> InstanceNorm backward chain on the main stream; under graph capture the
# fork / join become graph edges.  The join is inside the same backward, so autograd, GradSync hooks and the optimizer
# only ever see finished gradients on the main stream.  On the large levels the persistent one-workgroup-per-CU kernels
# cannot share the chip (measured in round 3: 17.2 vs 16.2 ms with everything on the side stream), so the switch is by
# size: RU3D_WGRAD_STREAM = 0 never, 1 always, unset: when a block's tensors have at most RU3D_SIDE_MAXVOX voxels in all
# (default 65536 = the 32^3 level at batch 2: 17.10 -> 16.92 ms same box; with the 16^3 / 8^3 levels alone 17.10 -> 17.11 -
# their kernels are bound by the CUs' L1 / LDS paths, which a concurrent kernel shares).
# End of round 4: OFF by default.  With the launch count down from 596 to 385 and the skip convs' weight gradients riding in
# their partners' kernels, the overlap no longer pays for itself: captured step 15.42 (auto) vs 15.38 ms (off), eager loop
# 15.63 vs 15.40 ms - the fork / join calls cost the host-bound eager loop more than the overlap returns (same box, three
# rounds each).  RU3D_WGRAD_STREAM=auto / 1 bring it back.
_SIDE_MODE = os.environ.get("RU3D_WGRAD_STREAM", "0")
_SIDE_MAXVOX = int(os.environ.get("RU3D_SIDE_MAXVOX", "65536"))
_SIDE = {}
_SIDE_BUSY = {}      # device -> tensors the side stream still reads (kept alive until the join)


def _side_stream(device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


class _OnSide:
    """`with _OnSide(device, nvox, keep):` - the body's launches are ordered after everything already on the current
    stream and run on the side stream; `_join()` makes the current stream wait for them.  keep: the body's input tensors
    (allocated on the main stream: they must not return to its allocator before the side stream has read them)."""

    def __init__(self, device, nvox=0, keep=()):
        self.device = device
        self.ctx = None
        self.on = _SIDE_MODE == "1" or (_SIDE_MODE == "auto" and 0 < nvox <= _SIDE_MAXVOX)
        self.keep = keep

    def __enter__(self):
        if self.on:
            side = _side_stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            _SIDE_BUSY.setdefault(self.device, []).extend(t for t in self.keep if t is not None)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def _join(device, *outs):
    busy = _SIDE_BUSY.get(device)
    if busy is None:
        return
    cur = torch.cuda.current_stream(device)
    cur.wait_stream(_side_stream(device))
    for t in outs:
        if t is not None:
            t.record_stream(cur)
    del _SIDE_BUSY[device]


# --------------------------------------------------------------------------- skip connection plumbing
class SkipLink:
    """One encoder -> decoder skip connection of a native U-Net (reference network.py:553-563: `skips.append(x)`,
    later `torch.cat((up, skip), dim=1)`).  Two copies disappear through it:
      * forward: the encoder block that produces the skip writes it straight into the second half of the buffer the
        decoder will use as its concat input (`buf`, allocated when the skip is produced), so UpFn has no channel copy
        to make;
      * backward: the skip tensor feeds the pooling block and the concat, and autograd would add the two gradients with
        an elementwise kernel; UpFn instead parks the concat's share here (`grad`) and the pooling ResBlock's last input
        gradient kernel adds it as its residual operand.  UpFn's backward always runs before the pooling block's (the
        decoder level sits above everything the pooling block feeds)."""

    def __init__(self, up_channels, dec_channels=0):
        self.up_channels = up_channels      # channels (padded when the net is padded) of the up-sampled half
        self.dec_channels = dec_channels    # output channels of the decoder block that consumes the concat
        self.buf = None
        self.grad = None
        self.planar = False                 # the concat is two planes of one buffer (N.Split), not interleaved channels

    def skip_view(self, n, c, d, h, w, dtype, device):
        self.planar = False
        self.buf = N.new_act(n, self.up_channels + c, d, h, w, dtype, device)
        return self.buf[:, self.up_channels:]

    def planar_view(self, n, c, d, h, w, dtype, device):
        """Full-resolution level (32 + 32 channels of 16 bits: interleaved, the halves would share every 128-byte line):
        the concat as two planes [up | skip] of one buffer when every kernel of the decoder block takes that
        (ru3d_planar_concat_supported); the skip plane, or None."""
        if (c != self.up_channels or dtype == torch.float32 or not self.dec_channels
                or not N.lib.ru3d_planar_concat_supported(n, d, h, w, c, self.dec_channels, N.dtype_code(dtype))):
            return None
        self.planar = True
        self.buf = N.new_act(2 * n, c, d, h, w, dtype, device)
        return self.buf[n:]


# --------------------------------------------------------------------------- autograd: plain conv (stem / head / skip)
def _f32_view(pack, count):
    """fp32 view of a ROLE_BIAS pack (the packs are 256-byte aligned slices of one uint8 allocation)."""
    return pack[:4 * count].view(torch.float32)


class ConvFn(torch.autograd.Function):
    """nn.Conv3d(k in {1,3}, stride in {1,2}, padding=k//2) with bias: reference network.py:541-547 (stem, head).
    pad_in / pad_out: the input / output activation carries its channels zero-padded to a multiple of 32."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, storage_dtype, out_dtype, pad_in=False, pad_out=False):
        xin = as_input(x, storage_dtype)
        k = weight.shape[2]
        cout, cin = weight.shape[0], weight.shape[1]
        cin_seg = seg_of(cin, xin.shape[1]) if pad_in else 0
        cout_seg = cout if pad_out else 0
        if not pad_in and xin.shape[1] != cin:
            raise N.Ru3dError("conv: input has %d channels, weight expects %d" % (xin.shape[1], cin))
        cout_p = padded_dim(cout, cout_seg)
        specs = [(weight, N.ROLE_CONV_FWD, stride, cout_seg, cin_seg)]
        if ctx.needs_input_grad[0]:
            specs.append((weight, N.ROLE_CONV_DGRAD, stride, cout_seg, cin_seg))
        if cout_seg and bias is not None:
            specs.append((bias, N.ROLE_BIAS, 1, cout_seg, 0))
        packs = pack_weights(specs, storage_dtype)
        b = _f32_view(packs[-1], cout_p) if (cout_seg and bias is not None) else bias
        y = conv_fwd(xin, packs[0], b, cout_p, k, stride, out_dtype=out_dtype)
        ctx.save_for_backward(xin, packs[1] if ctx.needs_input_grad[0] else None)
        ctx.dims = (cout, cin, cout_seg, cin_seg)
        ctx.stride, ctx.k, ctx.has_bias = stride, k, bias is not None
        ctx.storage_dtype = storage_dtype
        ctx.in_dtype = x.dtype
        ctx.wkey = weight.data_ptr() if not (cout_seg or cin_seg) else None
        ctx.weight_ref = weight if (k == 1 and cout <= 4) else None      # the head's fused backward reads the fp32 parameter
        return y

    @staticmethod
    def backward(ctx, gy):
        xin, pwd = ctx.saved_tensors
        sd = ctx.storage_dtype
        cout, cin, cout_seg, cin_seg = ctx.dims
        if (ctx.k == 1 and ctx.stride == 1 and cout <= 4 and not cout_seg and all(ctx.needs_input_grad[:2])
                and ctx.weight_ref is not None):
            # the head: input, weight and bias gradient from one pass over (x, dlogits)
            fused = head_bwd(xin, gy, ctx.weight_ref, ctx.has_bias and ctx.needs_input_grad[2],
                             key=None if cin_seg else ctx.wkey)
            if fused is not None:
                gx, gw, gb = fused
                if gx.dtype != ctx.in_dtype:
                    gx = gx.to(ctx.in_dtype)
                return gx, gw, gb, None, None, None, None, None
        gy = as_grad(gy, sd)
        gx = gw = gb = None
        with _OnSide(gy.device):      # stem / head: full-resolution tensors, main stream
            if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2] and not (cout_seg or cin_seg) \
                    and sd != torch.float32:
                gw, gb = conv_wgrad_bias(xin, gy, ctx.k, ctx.stride, key=ctx.wkey)       # the stem: db from the same pass
            elif ctx.needs_input_grad[1]:
                gw = unpad_wgrad(conv_wgrad(xin, gy, ctx.k, ctx.stride, key=ctx.wkey), cout, cin, cout_seg, cin_seg)
            if gb is None and ctx.has_bias and ctx.needs_input_grad[2]:
                gb = channel_sum(gy)[:cout]
        if ctx.needs_input_grad[0]:
            gx = conv_dgrad(gy, pwd, tuple(xin.shape), ctx.k, ctx.stride)
            if gx.dtype != ctx.in_dtype:
                gx = gx.to(ctx.in_dtype)
        _join(gy.device, gw, gb)
        return gx, gw, gb, None, None, None, None, None


# --------------------------------------------------------------------------- autograd: ResBlock
class ResBlockFn(torch.autograd.Function):
    """reference network.py:405-416:
        skip = skip_conv(x) if (in != out or stride != 1) else x
        x = conv1(x); x = dropout(x); x = lrelu(IN(x)); x = conv2(x); return lrelu(IN(x) + skip)
    pad: x and the result carry their channels zero-padded to multiples of 32 (F = 30 widths on the MFMA kernels).
    checkpoint: only the block's input, output and the InstanceNorm statistics are kept for backward; the three
    interior tensors (conv1 output, its activation, conv2 output) are recomputed there by the same kernels, so the
    gradients are the same bits (BASELINE config 5: activation checkpointing).
    """

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, bs, stride, drop_scale, pad=0, checkpoint=False, in_link=None,
                out_link=None, planar=False):
        """in_link: SkipLink whose parked gradient this block adds to its input gradient (pooling block);
        out_link: SkipLink into whose concat buffer the block writes its output (last encoder block of a level);
        planar: x is a split concat - the [2N, C/2, D, H, W] tensor of its two planes (N.Split, UpFn made it)."""
        sd = x.dtype
        x = N.to_ndhwc(x)
        xt = x                      # the tensor autograd sees
        if planar:
            x = N.Split(x)
        cout, cin = w1.shape[0], w1.shape[1]
        cin_seg = seg_of(cin, x.shape[1], int(pad)) if pad else 0      # pad = number of input segments
        cout_seg = cout if pad else 0
        if not pad and x.shape[1] != cin:
            raise N.Ru3dError("ResBlock: input has %d channels, conv1 expects %d" % (x.shape[1], cin))
        cout_p = padded_dim(cout, cout_seg)
        # every packed form this block needs (forward now, input gradients later) in one launch
        train = any(ctx.needs_input_grad)
        need_gx = ctx.needs_input_grad[0]
        specs = [(w1, N.ROLE_CONV_FWD, stride, cout_seg, cin_seg), (w2, N.ROLE_CONV_FWD, 1, cout_seg, cout_seg)]
        if ws is not None:
            specs.append((ws, N.ROLE_CONV_FWD, stride, cout_seg, cin_seg))
        nfwd = len(specs)
        if train:
            specs.append((w2, N.ROLE_CONV_DGRAD, 1, cout_seg, cout_seg))
            if need_gx:
                specs.append((w1, N.ROLE_CONV_DGRAD, stride, cout_seg, cin_seg))
                if ws is not None:
                    specs.append((ws, N.ROLE_CONV_DGRAD, stride, cout_seg, cin_seg))
        nw = len(specs)
        if cout_seg:
            specs += [(b, N.ROLE_BIAS, 1, cout_seg, 0) for b in (b1, b2, bs) if b is not None]
        packs = pack_weights(specs, sd)
        if cout_seg:
            it = iter(packs[nw:])
            b1, b2, bs = [(_f32_view(next(it), cout_p) if b is not None else None) for b in (b1, b2, bs)]
        pw1, pw2 = packs[0], packs[1]
        fused = conv_s2_pair_fwd_in(x, pw1, b1, packs[2], bs, cout_p, drop_scale) if (ws is not None and stride == 2) else None
        if fused is not None:       # pooling block: conv1 + its statistics and the skip conv from one read of x
            y1, mean1, scale1, skip = fused
            a1 = in_lrelu_fwd(y1, mean1, scale1)
        else:
            y1, mean1, scale1, a1 = conv_fwd_in_act(x, pw1, b1, cout_p, 3, stride, drop_scale)
        n_, _, d_, h_, w_ = a1.shape
        # in place only when a voxel's channels fill whole 128-byte lines of the interleaved [up | skip] buffer: at 32
        # 16-bit channels every reader of the skip (pool conv, its weight gradient, the norm backward) would pull the
        # other half's lines along - measured slower than the copy it saves
        zout = None
        if out_link is not None and cout_p * a1.element_size() >= 128:
            zout = out_link.skip_view(n_, cout_p, d_, h_, w_, sd, x.device)
        elif out_link is not None:
            zout = out_link.planar_view(n_, cout_p, d_, h_, w_, sd, x.device)      # None: the copy path
        z = None
        # decoder block on the large levels: skip conv + IN apply + sum + LeakyReLU in one pass, the skip never stored
        # (no link buffer then: decoder outputs are not skips); the shapes are skip1x1_fused_eligible's
        tail = (fused is None and ws is not None and stride == 1 and out_link is None and sd != torch.float32
                and (x.shape[1], cout_p) in ((64, 32), (128, 64)) and n_ * d_ * h_ * w_ >= 65536)
        if tail:
            y2, mean2, scale2 = conv_fwd_in(a1, pw2, b2, cout_p, 3, 1)
            z = skip1x1_in_lrelu_fwd(x, packs[2], bs, y2, mean2, scale2)
            if z is None:
                z = in_lrelu_fwd(y2, mean2, scale2, res=conv_fwd(x, packs[2], bs, cout_p, 1, stride), out=zout)
        else:
            if fused is not None:
                pass
            elif ws is not None:
                skip = conv_fwd(x, packs[2], bs, cout_p, 1, stride)
            else:
                skip = x
            y2, mean2, scale2, z = conv_fwd_in_act(a1, pw2, b2, cout_p, 3, 1, res=skip, out=zout)
        bwd = packs[nfwd:nw] + [None] * 3
        if checkpoint and train:
            ctx.save_for_backward(xt, None, None, None, z, mean1, scale1, mean2, scale2, bwd[0], bwd[1], bwd[2],
                                  pw1, pw2, b1, b2)
        else:
            ctx.save_for_backward(xt, y1, a1, y2, z, mean1, scale1, mean2, scale2, bwd[0], bwd[1], bwd[2],
                                  None, None, None, None)
        ctx.planar = bool(planar)
        ctx.dims = (cout, cin, cout_seg, cin_seg)
        ctx.stride = stride
        plain = not (cout_seg or cin_seg)
        ctx.wkeys = (w1.data_ptr() if plain else None, w2.data_ptr() if plain else None,
                     ws.data_ptr() if (plain and ws is not None) else None)
        ctx.has_skip_conv = ws is not None
        ctx.in_link = in_link if (in_link is not None and ws is not None) else None
        if in_link is not None:
            in_link.fused_grad = ctx.in_link is not None and need_gx      # UpFn parks its share only when it will be used
        return z

    @staticmethod
    def backward(ctx, gz):
        (x, y1, a1, y2, z, mean1, scale1, mean2, scale2, pw2d, pw1d, pwsd, pw1, pw2, b1, b2) = ctx.saved_tensors
        cout, cin, cout_seg, cin_seg = ctx.dims
        cout_p = padded_dim(cout, cout_seg)
        sd = x.dtype
        dev = x.device
        stride = ctx.stride
        gz = as_grad(gz, sd)
        if ctx.planar:
            x = N.Split(x)
        if y1 is None:      # checkpointed: the same kernels on the same inputs give the same bits
            y1 = conv_fwd(x, pw1, b1, cout_p, 3, stride)
            a1 = in_lrelu_fwd(y1, mean1, scale1)
            y2 = conv_fwd(a1, pw2, b2, cout_p, 3, 1)
        # lrelu(IN(y2) + skip): dy2 and the pre-activation gradient (= d/dskip)
        # sum(gpre) - the skip conv's bias gradient - comes out of the same reduction
        dy2, gpre, gbs_sum = in_lrelu_bwd(gz, z, y2, mean2, scale2, want_gpre=True, want_gpre_sum=True)
        del y2
        return ResBlockFn._backward_tail(ctx, x, a1, y1, dy2, gpre, gbs_sum, mean1, scale1, pw2d, pw1d, pwsd)

    @staticmethod
    def _backward_tail(ctx, x, a1, y1, dy2, gpre, gbs_sum, mean1, scale1, pw2d, pw1d, pwsd):
        cout, cin, cout_seg, cin_seg = ctx.dims
        dev = x.device
        stride = ctx.stride
        gws = gbs = None
        nvox = dy2.shape[0] * dy2.shape[2] * dy2.shape[3] * dy2.shape[4]
        xk = x.t if ctx.planar else x
        # a block with a skip conv (decoder blocks: stride 1, pooling blocks: stride 2): its weight gradient rides in the
        # free tap slot of conv1's weight-gradient kernel further down - one pass over x instead of two
        pair_later = ctx.has_skip_conv and _wgrad_pair_ok(x, gpre, stride)
        with _OnSide(dev, nvox, (a1, dy2, xk, gpre)):
            gw2 = unpad_wgrad(conv_wgrad(a1, dy2, 3, 1, key=ctx.wkeys[1]), cout, cout, cout_seg, cout_seg)
            if ctx.has_skip_conv:
                if not pair_later:
                    gws = unpad_wgrad(conv_wgrad(x, gpre, 1, stride, key=ctx.wkeys[2]), cout, cin, cout_seg, cin_seg)
                gbs = gbs_sum[:cout]
        gb2 = None   # a bias that feeds InstanceNorm has an identically zero gradient: reported as "no gradient"
        # conv2's input gradient and the IN1 + LeakyReLU backward in one call: on the sliding-kernel shapes the backward's
        # sums are taken in the conv's epilogue
        dy1 = conv_dgrad_in_bwd(dy2, pw2d, a1, mean1, scale1)
        del dy2, a1, y1
        with _OnSide(dev, nvox, (xk, dy1, gpre)):
            both = conv_wgrad_pair(x, dy1, gpre, stride, key=ctx.wkeys[0], key2=ctx.wkeys[2]) if pair_later else None
            if both is not None:
                gw1 = unpad_wgrad(both[0], cout, cin, cout_seg, cin_seg)
                gws = unpad_wgrad(both[1], cout, cin, cout_seg, cin_seg)
            else:
                gw1 = unpad_wgrad(conv_wgrad(x, dy1, 3, stride, key=ctx.wkeys[0]), cout, cin, cout_seg, cin_seg)
                if pair_later:
                    gws = unpad_wgrad(conv_wgrad(x, gpre, 1, stride, key=ctx.wkeys[2]), cout, cin, cout_seg, cin_seg)
        gb1 = None
        gx = None
        need_gx = ctx.needs_input_grad[0]
        if ctx.has_skip_conv:
            if need_gx:
                parked = None
                if ctx.in_link is not None:
                    parked, ctx.in_link.grad = ctx.in_link.grad, None
                if stride == 2:     # both stride-2 input gradients (+ the parked concat share) in one launch
                    gx = conv_s2_dgrad_pair(dy1, pw1d, gpre, pwsd, tuple(x.shape), res=parked)
                elif parked is None:
                    gx = conv_s1_dgrad_pair(dy1, pw1d, gpre, pwsd, tuple(x.shape), planar=ctx.planar)
                if gx is None and ctx.planar:
                    raise N.Ru3dError("ResBlock: the split concat input has no fused input-gradient kernel for this shape")
                if gx is None:
                    gx0 = conv_dgrad(gpre, pwsd, tuple(x.shape), 1, stride, res=parked)
                    gx = conv_dgrad(dy1, pw1d, tuple(x.shape), 3, stride, res=gx0)
        elif need_gx:
            gx = conv_dgrad(dy1, pw1d, tuple(x.shape), 3, stride, res=gpre)
        _join(dev, gw1, gw2, gws, gbs)
        return gx, gw1, gb1, gw2, gb2, gws, gbs, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- autograd: ConvTrans3D (+ concat)
class UpFn(torch.autograd.Function):
    """reference network.py:311-317 (+ :346-350 when `skip` is given):
        u = lrelu(IN(pad_far(convT_k3s2p1(x))));  return cat((u, skip), dim=1)
    The concat is written in place: the IN+LeakyReLU kernel stores into the first channels of the
    output buffer, a channel-slice copy fills the rest.  pad: channel-padded activations (see ResBlockFn); the
    concat is then [u padded | skip padded].
    """

    @staticmethod
    def forward(ctx, x, wt, bt, skip, pad=False, link=None):
        sd = x.dtype
        x = N.to_ndhwc(x)
        cin, cout = wt.shape[0], wt.shape[1]
        cin_seg = seg_of(cin, x.shape[1]) if pad else 0
        cout_seg = cout if pad else 0
        if not pad and x.shape[1] != cin:
            raise N.Ru3dError("ConvTrans3D: input has %d channels, weight expects %d" % (x.shape[1], cin))
        cout_p = padded_dim(cout, cout_seg)
        specs = [(wt, N.ROLE_CONVT_FWD, 2, cout_seg, cin_seg)]
        if ctx.needs_input_grad[0]:
            specs.append((wt, N.ROLE_CONVT_DGRAD, 2, cout_seg, cin_seg))
        if cout_seg and bt is not None:
            specs.append((bt, N.ROLE_BIAS, 1, cout_seg, 0))
        packs = pack_weights(specs, sd)
        btp = _f32_view(packs[-1], cout_p) if (cout_seg and bt is not None) else bt
        y, mean, scale = convt_fwd_in(x, packs[0], btp, cout_p)
        n, _, d, h, w = y.shape
        if skip is not None:
            skip = as_grad(skip, sd)
            cs = skip.shape[1]
            if tuple(skip.shape[2:]) != (d, h, w) or skip.shape[0] != n:
                raise N.Ru3dError("UpConcat: skip %s does not match up-sampled %s" % (tuple(skip.shape), tuple(y.shape)))
            planar = (link is not None and link.planar and link.buf is not None and link.up_channels == cout_p == cs
                      and tuple(link.buf.shape) == (2 * n, cs, d, h, w) and link.buf.dtype == sd
                      and skip.data_ptr() == link.buf[n:].data_ptr())
            in_place = (not planar and link is not None and not link.planar and link.buf is not None
                        and link.up_channels == cout_p
                        and tuple(link.buf.shape) == (n, cout_p + cs, d, h, w) and link.buf.dtype == sd
                        and skip.data_ptr() == link.buf[:, cout_p:].data_ptr())
            if planar:
                # the concat as two planes [u | skip] of one buffer: the encoder wrote the skip plane, u is a dense tensor
                # of its own; the decoder block takes the pair as a split tensor (N.Split)
                buf = link.buf
                in_lrelu_fwd(y, mean, scale, out=buf[:n])
            else:
                buf = link.buf if in_place else N.new_act(n, cout_p + cs, d, h, w, sd, x.device)
                u = buf[:, :cout_p]
                in_lrelu_fwd(y, mean, scale, out=u)
                if not in_place:
                    copy_channels(skip, buf[:, cout_p:])
            if link is not None:
                link.buf = None          # the autograd graph owns the buffer from here on
                link.planar_out = planar
            out = buf
        else:
            u = in_lrelu_fwd(y, mean, scale)
            out = u
        ctx.save_for_backward(x, y, out, mean, scale, packs[1] if ctx.needs_input_grad[0] else None)
        ctx.dims = (cout, cin, cout_seg, cin_seg)
        ctx.has_skip = skip is not None
        ctx.planar = bool(skip is not None and link is not None and getattr(link, "planar_out", False))
        ctx.wkey = wt.data_ptr() if not (cout_seg or cin_seg) else None
        ctx.link = link if (link is not None and getattr(link, "fused_grad", False)) else None
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, out, mean, scale, pwd = ctx.saved_tensors
        cout, cin, cout_seg, cin_seg = ctx.dims
        cout_p = padded_dim(cout, cout_seg)
        sd = x.dtype
        g = as_grad(g, sd)
        if ctx.planar:              # planes [u | skip] of one buffer: dense halves
            n = out.shape[0] // 2
            u, gu, gskip = out[:n], g[:n], g[n:]
        else:
            u = out[:, :cout_p] if ctx.has_skip else out
            gu = g[:, :cout_p] if ctx.has_skip else g
            gskip = g[:, cout_p:] if ctx.has_skip else None
        gb = torch.empty(y.shape[1], dtype=torch.float32, device=y.device)
        dy, _ = in_lrelu_bwd(gu, u, y, mean, scale, zero_far=True, dy_sum=gb)       # bias gradient: sums of dy
        gb = gb[:cout]
        nvox = x.shape[0] * x.shape[2] * x.shape[3] * x.shape[4]
        with _OnSide(x.device, nvox, (x, dy)):
            # the ConvTranspose3d weight is [Cin][Cout][27]: its outer dimension is the module's in_channels
            gw = unpad_wgrad(convt_wgrad(x, dy, key=ctx.wkey), cin, cout, cin_seg, cout_seg)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = convt_dgrad(dy, pwd, tuple(x.shape))
        _join(x.device, gw, gb)
        if ctx.link is not None and gskip is not None:
            ctx.link.grad, gskip = gskip, None       # added by the pooling block's input-gradient kernel (SkipLink)
        return gx, gw, gb, gskip, None, None


# --------------------------------------------------------------------------- autograd: BatchNorm blocks, training mode
class ResBlockBNFn(torch.autograd.Function):
    """ResBlock built with norm_op=nn.BatchNorm3d (reference network.py:38-69, 405-416) in training mode:
        skip = skip_conv(x) | x;  x = conv1(x); x = dropout(x); x = lrelu(BN(x)); x = conv2(x); lrelu(BN(x) + skip)
    with the block's ONE BatchNorm3d module used twice (network.py:401: both calls move the running averages, in this
    order; gamma / beta gradients are the sum of the two uses).  Same kernels as ResBlockFn around the batch-pooled
    statistics (bn_train_stats / bn_lrelu_bwd); `norm` is the module (running buffers are updated in place)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, bs, gamma, beta, stride, drop_scale, pad, norm):
        sd = x.dtype
        x = N.to_ndhwc(x)
        cout, cin = w1.shape[0], w1.shape[1]
        cin_seg = seg_of(cin, x.shape[1], int(pad)) if pad else 0
        cout_seg = cout if pad else 0
        if not pad and x.shape[1] != cin:
            raise N.Ru3dError("ResBlock: input has %d channels, conv1 expects %d" % (x.shape[1], cin))
        cout_p = padded_dim(cout, cout_seg)
        need_gx = ctx.needs_input_grad[0]
        specs = [(w1, N.ROLE_CONV_FWD, stride, cout_seg, cin_seg), (w2, N.ROLE_CONV_FWD, 1, cout_seg, cout_seg)]
        if ws is not None:
            specs.append((ws, N.ROLE_CONV_FWD, stride, cout_seg, cin_seg))
        nfwd = len(specs)
        specs.append((w2, N.ROLE_CONV_DGRAD, 1, cout_seg, cout_seg))
        if need_gx:
            specs.append((w1, N.ROLE_CONV_DGRAD, stride, cout_seg, cin_seg))
            if ws is not None:
                specs.append((ws, N.ROLE_CONV_DGRAD, stride, cout_seg, cin_seg))
        nw = len(specs)
        if cout_seg:
            specs += [(b, N.ROLE_BIAS, 1, cout_seg, 0) for b in (b1, b2, bs) if b is not None]
        packs = pack_weights(specs, sd)
        has_b = (b1 is not None, b2 is not None, bs is not None)
        if cout_seg:
            it = iter(packs[nw:])
            b1, b2, bs = [(_f32_view(next(it), cout_p) if b is not None else None) for b in (b1, b2, bs)]
        y1 = conv_fwd(x, packs[0], b1, cout_p, 3, stride)
        fs1, fh1, a1s, b1s, cnt1 = bn_train_stats(y1, drop_scale, norm, cout)
        a1 = affine_lrelu_fwd(y1, fs1, fh1)
        y2 = conv_fwd(a1, packs[1], b2, cout_p, 3, 1)
        fs2, fh2, a2s, b2s, cnt2 = bn_train_stats(y2, None, norm, cout)
        skip = conv_fwd(x, packs[2], bs, cout_p, 1, stride) if ws is not None else x
        z = affine_lrelu_fwd(y2, fs2, fh2, res=skip)
        bwd = packs[nfwd:nw] + [None] * 3
        ctx.save_for_backward(x, y1, a1, y2, z, fs1, a1s, b1s, fs2, a2s, b2s, bwd[0], bwd[1], bwd[2])
        ctx.dims = (cout, cin, cout_seg, cin_seg)
        ctx.counts = (cnt1, cnt2)
        ctx.stride = stride
        ctx.has_skip_conv = ws is not None
        ctx.has_b = has_b
        return z

    @staticmethod
    def backward(ctx, gz):
        x, y1, a1, y2, z, fs1, a1s, b1s, fs2, a2s, b2s, pw2d, pw1d, pwsd = ctx.saved_tensors
        cout, cin, cout_seg, cin_seg = ctx.dims
        stride = ctx.stride
        gz = as_grad(gz, x.dtype)
        dy2, gpre, dg2, db2 = bn_lrelu_bwd(gz, z, y2, a2s, b2s, fs2, ctx.counts[1])
        gw2 = unpad_wgrad(conv_wgrad(a1, dy2, 3, 1), cout, cout, cout_seg, cout_seg)
        gb2 = channel_sum(dy2)[:cout] if ctx.has_b[1] else None
        gws = gbs = None
        if ctx.has_skip_conv:
            gws = unpad_wgrad(conv_wgrad(x, gpre, 1, stride), cout, cin, cout_seg, cin_seg)
            gbs = db2[:cout].clone() if ctx.has_b[2] else None        # sum of gpre over n and voxels
        da1 = conv_dgrad(dy2, pw2d, tuple(a1.shape), 3, 1)
        del dy2
        dy1, _, dg1, db1 = bn_lrelu_bwd(da1, a1, y1, a1s, b1s, fs1, ctx.counts[0])
        del da1
        gw1 = unpad_wgrad(conv_wgrad(x, dy1, 3, stride), cout, cin, cout_seg, cin_seg)
        # with Dropout3d between conv1 and a BATCH norm the bias does not cancel (its share d[n][c] * b differs per sample)
        gb1 = channel_sum(dy1)[:cout] if ctx.has_b[0] else None
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.has_skip_conv:
                gx0 = conv_dgrad(gpre, pwsd, tuple(x.shape), 1, stride)
                gx = conv_dgrad(dy1, pw1d, tuple(x.shape), 3, stride, res=gx0)
            else:
                gx = conv_dgrad(dy1, pw1d, tuple(x.shape), 3, stride, res=gpre)
        ggamma = (dg1 + dg2)[:cout] if ctx.needs_input_grad[7] else None
        gbeta = (db1 + db2)[:cout] if ctx.needs_input_grad[8] else None
        return gx, gw1, gb1, gw2, gb2, gws, gbs, ggamma, gbeta, None, None, None, None


class UpBNFn(torch.autograd.Function):
    """ConvTrans3D built with norm_op=nn.BatchNorm3d in training mode (reference network.py:311-317, + :346-350 when
    `skip` is given): u = lrelu(BN(pad_far(convT_k3s2p1(x)))); return cat((u, skip), dim=1).  The zero far planes are
    part of the batch statistics, as in the reference (the pad sits in front of the norm)."""

    @staticmethod
    def forward(ctx, x, wt, bt, gamma, beta, skip, pad, norm):
        sd = x.dtype
        x = N.to_ndhwc(x)
        cin, cout = wt.shape[0], wt.shape[1]
        cin_seg = seg_of(cin, x.shape[1]) if pad else 0
        cout_seg = cout if pad else 0
        if not pad and x.shape[1] != cin:
            raise N.Ru3dError("ConvTrans3D: input has %d channels, weight expects %d" % (x.shape[1], cin))
        cout_p = padded_dim(cout, cout_seg)
        specs = [(wt, N.ROLE_CONVT_FWD, 2, cout_seg, cin_seg)]
        if ctx.needs_input_grad[0]:
            specs.append((wt, N.ROLE_CONVT_DGRAD, 2, cout_seg, cin_seg))
        if cout_seg and bt is not None:
            specs.append((bt, N.ROLE_BIAS, 1, cout_seg, 0))
        packs = pack_weights(specs, sd)
        btp = _f32_view(packs[-1], cout_p) if (cout_seg and bt is not None) else bt
        y = convt_fwd(x, packs[0], btp, cout_p)
        fs, fh, a_s, b_s, cnt = bn_train_stats(y, None, norm, cout)
        n, _, d, h, w = y.shape
        if skip is not None:
            skip = as_grad(skip, sd)
            cs = skip.shape[1]
            if tuple(skip.shape[2:]) != (d, h, w) or skip.shape[0] != n:
                raise N.Ru3dError("UpConcat: skip %s does not match up-sampled %s" % (tuple(skip.shape), tuple(y.shape)))
            out = N.new_act(n, cout_p + cs, d, h, w, sd, x.device)
            affine_lrelu_fwd(y, fs, fh, out=out[:, :cout_p])
            copy_channels(skip, out[:, cout_p:])
        else:
            out = affine_lrelu_fwd(y, fs, fh)
        ctx.save_for_backward(x, y, out, fs, a_s, b_s, packs[1] if ctx.needs_input_grad[0] else None)
        ctx.dims = (cout, cin, cout_seg, cin_seg)
        ctx.count = cnt
        ctx.has_skip = skip is not None
        ctx.has_b = bt is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, out, fs, a_s, b_s, pwd = ctx.saved_tensors
        cout, cin, cout_seg, cin_seg = ctx.dims
        cout_p = padded_dim(cout, cout_seg)
        g = as_grad(g, x.dtype)
        u = out[:, :cout_p] if ctx.has_skip else out
        gu = g[:, :cout_p] if ctx.has_skip else g
        gskip = g[:, cout_p:] if ctx.has_skip else None
        dy, _, dgam, dbet = bn_lrelu_bwd(gu, u, y, a_s, b_s, fs, ctx.count, zero_far=True)
        gw = unpad_wgrad(convt_wgrad(x, dy), cin, cout, cin_seg, cout_seg)
        gb = channel_sum(dy)[:cout] if ctx.has_b else None
        gx = convt_dgrad(dy, pwd, tuple(x.shape)) if ctx.needs_input_grad[0] else None
        return (gx, gw, gb, dgam[:cout] if ctx.needs_input_grad[3] else None,
                dbet[:cout] if ctx.needs_input_grad[4] else None, gskip, None, None)


# --------------------------------------------------------------------------- autograd: attention gate (+ concat)
class AttGateFn(torch.autograd.Function):
    """reference network.py:365-371 (AttBlock.forward) followed by network.py:350 (cat((up, gated_skip), dim=1)):
        x = conv(skip); g = conv(up); rate = sigmoid(conv(lrelu(x + g))); out = cat((up, x * rate))
    with ONE shared 1x1x1 convolution (weight [C, C, 1, 1, 1], bias).  The three convolutions and their gradients
    run on the conv kernels, the elementwise pieces on ru3d_pointwise; the weight gradient is the sum of the three
    uses.  pad: channel-padded activations (see ResBlockFn)."""

    @staticmethod
    def forward(ctx, skip, up, w, b, pad=False):
        sd = up.dtype
        skip = as_grad(skip, sd)
        up = N.to_ndhwc(up)
        c = w.shape[0]
        seg = c if pad else 0
        cp = padded_dim(c, seg)
        if skip.shape[1] != cp or up.shape[1] != cp:
            raise N.Ru3dError("AttBlock: skip %s / gate %s do not carry %d channels" % (tuple(skip.shape), tuple(up.shape), cp))
        specs = [(w, N.ROLE_CONV_FWD, 1, seg, seg), (w, N.ROLE_CONV_DGRAD, 1, seg, seg)]
        if seg and b is not None:
            specs.append((b, N.ROLE_BIAS, 1, seg, 0))
        packs = pack_weights(specs, sd)
        bp = _f32_view(packs[-1], cp) if (seg and b is not None) else b
        xs = conv_fwd(skip, packs[0], bp, cp, 1, 1)
        t = conv_fwd(up, packs[0], bp, cp, 1, 1, res=xs)          # conv(up) + b + x
        f = pointwise(0, t)
        del t
        r = conv_fwd(f, packs[0], bp, cp, 1, 1)
        gated = pointwise(1, xs, r)
        n, _, d, h, wd = up.shape
        out = N.new_act(n, 2 * cp, d, h, wd, sd, up.device)
        copy_channels(up, out[:, :cp])
        copy_channels(gated, out[:, cp:])
        ctx.save_for_backward(skip, up, xs, f, r, packs[1])
        ctx.dims = (c, seg, cp, b is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        skip, up, xs, f, r, pwd = ctx.saved_tensors
        c, seg, cp, has_bias = ctx.dims
        sd = up.dtype
        g = as_grad(g, sd)
        g_up_direct, g_gated = g[:, :cp], g[:, cp:]
        dxs1, dr = pointwise(2, xs, r, g_gated, n_out=2)
        d_f = conv_dgrad(dr, pwd, tuple(f.shape), 1, 1)
        d_t, d_xs = pointwise(3, f, dxs1, d_f, n_out=2)              # d_t = d_f * lrelu'(f); d_xs = d_t + dxs1
        gw = conv_wgrad(f, dr, 1, 1) + conv_wgrad(up, d_t, 1, 1) + conv_wgrad(skip, d_xs, 1, 1)
        gw = unpad_wgrad(gw, c, c, seg, seg)
        gb = None
        if has_bias:
            gb = (channel_sum(dr) + channel_sum(d_t) + channel_sum(d_xs))[:c]
        g_up = conv_dgrad(d_t, pwd, tuple(up.shape), 1, 1, res=g_up_direct)
        g_skip = conv_dgrad(d_xs, pwd, tuple(skip.shape), 1, 1)
        return g_skip, g_up, gw, gb, None
