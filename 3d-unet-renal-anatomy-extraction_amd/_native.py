"""ctypes binding of libru3d.so (the C ABI declared in include/ru3d.h).

There is NO fallback: if the shared library is missing the import fails loudly, and every op
raises when handed a tensor that does not live on a HIP device.  PyTorch is used only for
device memory, streams and autograd bookkeeping.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RU3D_LIB", os.path.join(_HERE, "libru3d.so"))

F32, BF16, F16 = 0, 1, 2
LABEL_I64, LABEL_U8 = 0, 1
ROLE_CONV_FWD, ROLE_CONV_DGRAD, ROLE_CONVT_FWD, ROLE_CONVT_DGRAD, ROLE_BIAS = 0, 1, 2, 3, 4
LOSS_HYBIRD, LOSS_DICELOSS, LOSS_FOCAL, LOSS_DICE = 0, 1, 2, 3
MAX_CLASSES = 8


class Tensor(ctypes.Structure):
    """struct ru3d_tensor"""
    _fields_ = [("ptr", ctypes.c_void_p), ("n", ctypes.c_int32), ("d", ctypes.c_int32), ("h", ctypes.c_int32),
                ("w", ctypes.c_int32), ("c", ctypes.c_int32), ("ld", ctypes.c_int32), ("cseg", ctypes.c_int32),
                ("seg_stride", ctypes.c_int64)]


class PackItem(ctypes.Structure):
    """struct ru3d_pack_item"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("cout", ctypes.c_int32), ("cin", ctypes.c_int32),
                ("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("role", ctypes.c_int32),
                ("cout_seg", ctypes.c_int32), ("cin_seg", ctypes.c_int32)]


class PatchParams(ctypes.Structure):
    """struct ru3d_patch_params"""
    _fields_ = [("lo", ctypes.c_int32 * 3), ("before", ctypes.c_int32 * 3), ("patch", ctypes.c_int32 * 3),
                ("flip", ctypes.c_int32 * 3), ("image_cval", ctypes.c_float), ("label_cval", ctypes.c_int32),
                ("do_contrast", ctypes.c_int32), ("do_brightness", ctypes.c_int32), ("do_gamma", ctypes.c_int32),
                ("contrast", ctypes.c_float), ("brightness", ctypes.c_float), ("gamma", ctypes.c_float),
                ("gamma_eps", ctypes.c_float)]


PACK_MAX = 40
_P = ctypes.POINTER(Tensor)
_vp, _i, _i64, _f, _sz, _u64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t,
                                ctypes.c_uint64)
_dbl = ctypes.c_double

# name -> (restype, argtypes): every symbol include/ru3d.h declares
SIGNATURES = {
    "ru3d_version": (_i, []),
    "ru3d_last_error": (ctypes.c_char_p, []),
    "ru3d_packed_weight_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ru3d_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ru3d_pack_weights": (_i, [ctypes.POINTER(PackItem), _i, _i, _vp]),
    "ru3d_unpad_weight_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ru3d_conv3d_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_fwd": (_i, [_P, _vp, _vp, _P, _P, _i, _i, _i, _i, _vp, _sz, _vp]),
    "ru3d_conv3d_fwd_in_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_fwd_in": (_i, [_P, _vp, _vp, _P, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _f, _vp]),
    "ru3d_conv3d_fwd_in_lrelu_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_fwd_in_lrelu": (_i, [_P, _vp, _vp, _P, _i, _i, _i, _vp, _vp, _vp, _P, _P, _f, _vp, _sz, _f, _vp]),
    "ru3d_planar_concat_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "ru3d_conv3d_dgrad": (_i, [_P, _vp, _P, _P, _i, _i, _i, _vp, _sz, _vp]),
    "ru3d_conv3d_s2_pair_fwd_in_supported": (_i, [_P, _P, _P, _i]),
    "ru3d_conv3d_s2_pair_fwd_in_workspace_bytes": (_sz, [_P, _P, _i]),
    "ru3d_conv3d_s2_pair_fwd_in": (_i, [_P, _vp, _vp, _P, _vp, _vp, _P, _vp, _vp, _vp, _vp, _sz, _f, _i, _vp]),
    "ru3d_conv3d_s1_dgrad_pair_supported": (_i, [_P, _P, _P, _i]),
    "ru3d_conv3d_s1_dgrad_pair": (_i, [_P, _vp, _P, _vp, _P, _i, _vp]),
    "ru3d_conv3d_s2_dgrad_pair_supported": (_i, [_P, _P, _P, _P, _i]),
    "ru3d_conv3d_s2_dgrad_pair": (_i, [_P, _vp, _P, _vp, _P, _P, _i, _vp]),
    "ru3d_conv3d_wgrad_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_wgrad": (_i, [_P, _P, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "ru3d_conv3d_wgrad_bias_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_wgrad_bias": (_i, [_P, _P, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "ru3d_conv3d_wgrad_pair_supported": (_i, [_P, _P, _P, _i, _i]),
    "ru3d_conv3d_wgrad_pair_workspace_bytes": (_sz, [_P, _P, _P, _i, _i]),
    "ru3d_conv3d_wgrad_pair": (_i, [_P, _P, _P, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "ru3d_head_bwd_supported": (_i, [_P, _P, _P, _i]),
    "ru3d_head_bwd_workspace_bytes": (_sz, [_P, _i]),
    "ru3d_head_bwd": (_i, [_P, _P, _vp, _i, _P, _vp, _vp, _vp, _sz, _i, _vp]),
    "ru3d_convtranspose3d_k3s2p1_fwd": (_i, [_P, _vp, _vp, _P, _i, _vp]),
    "ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes": (_sz, [_P, _P, _i]),
    "ru3d_convtranspose3d_k3s2p1_fwd_in": (_i, [_P, _vp, _vp, _P, _vp, _vp, _vp, _sz, _f, _i, _vp]),
    "ru3d_convtranspose3d_k3s2p1_dgrad": (_i, [_P, _vp, _P, _i, _vp]),
    "ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes": (_sz, [_P, _P, _i]),
    "ru3d_convtranspose3d_k3s2p1_wgrad": (_i, [_P, _P, _vp, _vp, _sz, _i, _vp]),
    "ru3d_reduce_workspace_bytes": (_sz, [_P]),
    "ru3d_instnorm_stats": (_i, [_P, _vp, _vp, _vp, _vp, _sz, _f, _i, _vp]),
    "ru3d_in_lrelu_fwd": (_i, [_P, _vp, _vp, _P, _P, _f, _i, _vp]),
    "ru3d_skip1x1_in_lrelu_fwd_supported": (_i, [_P, _P, _P, _i]),
    "ru3d_skip1x1_in_lrelu_fwd": (_i, [_P, _vp, _vp, _P, _vp, _vp, _P, _f, _i, _vp]),
    "ru3d_in_lrelu_bwd": (_i, [_P, _P, _P, _vp, _vp, _P, _P, _vp, _sz, _f, _i, _vp, _vp, _i, _vp]),
    "ru3d_in_lrelu_bwd_apply": (_i, [_P, _P, _vp, _vp, _vp, _P, _f, _i, _i, _vp]),
    "ru3d_conv3d_dgrad_in_bwd_workspace_bytes": (_sz, [_P, _P, _i, _i, _i]),
    "ru3d_conv3d_dgrad_in_bwd": (_i, [_P, _vp, _P, _vp, _vp, _P, _P, _i, _i, _f, _i, _vp, _sz, _vp]),
    "ru3d_channel_sum": (_i, [_P, _vp, _vp, _sz, _i, _vp]),
    "ru3d_batchnorm_stats_pool": (_i, [_P, _vp, _vp, _vp, _sz, _i, _vp]),
    "ru3d_batchnorm_stats_finalize": (_i, [_vp, _i, _i, _i, _dbl, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ru3d_affine_lrelu_fwd": (_i, [_P, _vp, _vp, _P, _P, _f, _i, _vp]),
    "ru3d_batchnorm_bwd_pool": (_i, [_P, _P, _P, _vp, _vp, _P, _vp, _vp, _sz, _f, _i, _vp]),
    "ru3d_batchnorm_bwd_apply": (_i, [_P, _P, _vp, _vp, _vp, _vp, _dbl, _P, _vp, _sz, _i, _i, _vp]),
    "ru3d_dropout3d_scale": (_i, [_vp, _i, _f, _u64, _u64, _vp]),
    "ru3d_dropout3d_scale_dev": (_i, [_vp, _i, _f, _u64, _u64, _vp, _vp]),
    "ru3d_pointwise": (_i, [_i, _P, _P, _P, _P, _P, _f, _i, _vp]),
    "ru3d_copy_channels": (_i, [_P, _P, _i, _vp]),
    "ru3d_add": (_i, [_P, _P, _P, _i, _vp]),
    "ru3d_cast_f32": (_i, [_P, _P, _i, _vp]),
    "ru3d_ncdhw_to_ndhwc": (_i, [_vp, _P, _i, _vp]),
    "ru3d_ndhwc_to_ncdhw": (_i, [_P, _vp, _i, _vp]),
    "ru3d_loss_state_bytes": (_sz, [_i]),
    "ru3d_loss_state_bad_labels_offset": (_sz, []),
    "ru3d_loss_workspace_bytes": (_sz, [_i, _i64, _i]),
    "ru3d_loss_fwd": (_i, [_vp, _i64, _i64, _i64, _vp, _i, _i, _i64, _i, _i, _f, _vp, _f, _f, _f, _vp, _vp, _vp,
                           _sz, _vp]),
    "ru3d_loss_bwd": (_i, [_vp, _i64, _i64, _i64, _vp, _i, _i, _i64, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "ru3d_tversky": (_i, [_vp, _vp, _i64, _f, _f, _f, _vp, _vp, _sz, _vp]),
    "ru3d_predict_accumulate": (_i, [_P, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ru3d_predict_merge": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ru3d_adam_multi": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "ru3d_adam_multi_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ru3d_adam_multi_amp": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ru3d_amp_update": (_i, [_vp, _f, _f, _i, _f, _f, _vp]),
    "ru3d_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "ru3d_augment_workspace_bytes": (_sz, [_i, _i, _i]),
    "ru3d_augment_label_presence": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int32), _i, _vp, _vp]),
    "ru3d_augment_patch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, ctypes.POINTER(PatchParams), _vp, _vp, _vp, _vp, _sz,
                               _vp]),
    "ru3d_grad_scale_check": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp]),
    "ru3d_comm_unique_id": (_i, [_vp]),
    "ru3d_comm_init": (_i, [ctypes.POINTER(_vp), _vp, _i, _i, _i]),
    "ru3d_comm_allreduce": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "ru3d_comm_reduce_scatter": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "ru3d_comm_all_gather": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ru3d_comm_available": (_i, []),
    "ru3d_comm_destroy": (_i, [_vp]),
    "ru3d_probe_begin": (_i, [_i, _i, _i, _i, _i, _i]),
    "ru3d_probe_end": (_i, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)]),
    "ru3d_set_cu_budget_device": (_i, [_i, _i]),
    "ru3d_set_cu_budget": (_i, [_i]),
    "ru3d_get_cu_budget": (_i, []),
    "ru3d_flat_cast": (_i, [_vp, _i, _vp, _i, _i64, _f, _vp]),
}
COMM_ID_BYTES = 128


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libru3d.so not found at %s - build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no non-HIP fallback for the 3D U-Net hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class Ru3dError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib.ru3d_last_error().decode("utf-8", "replace")
        raise Ru3dError("ru3d %s failed (status %d): %s" % (what, rc, msg))


_tls = threading.local()


def note_device(device):
    """Remember the device of the operands of the entry point being assembled (per thread: autograd runs backward
    on its own threads)."""
    _tls.device = device


def stream(device=None):
    """HIP stream handed to the C ABI: torch's current stream OF THE OPERANDS' DEVICE (not of whatever device the
    calling thread has current).  The library makes that stream's device current for the launch.  device: the
    operands' device when the caller has it at hand; otherwise the one the last desc() / note_device() /
    require_device() of this thread noted - ptr() checks every operand against it, so a stale note cannot send one
    GPU's pointers to another GPU's stream."""
    if device is not None:
        _tls.device = device
    dev = getattr(_tls, "device", None)
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise Ru3dError("ru3d: unsupported storage dtype %s (float32, bfloat16 or float16)" % dt)


def require_device(t, what="tensor"):
    if not t.is_cuda:
        raise Ru3dError("ru3d: %s lives on %s; the MI355X-native path has no CPU fallback - move the model "
                        "and its inputs to a HIP device (.cuda())" % (what, t.device))
    _tls.device = t.device


# --------------------------------------------------------------------------- activations
def new_act(n, c, d, h, w, dtype, device, zero=False):
    """[N,C,D,H,W]-shaped tensor whose memory is NDHWC (torch.channels_last_3d strides)."""
    buf = (torch.zeros if zero else torch.empty)((n, d, h, w, c), dtype=dtype, device=device)
    return buf.permute(0, 4, 1, 2, 3)


def is_ndhwc(t):
    """True when t ([N,C,D,H,W]) is dense NDHWC with an optional channel pitch (a channel slice)."""
    if t.dim() != 5:
        return False
    n, c, d, h, w = t.shape
    sn, sc, sd, sh, sw = t.stride()
    if c > 1 and sc != 1:
        return False
    ld = None
    for size, stride, inner in ((w, sw, 1), (h, sh, w), (d, sd, h * w), (n, sn, d * h * w)):
        if size > 1:
            if stride % inner != 0:
                return False
            cand = stride // inner
            if ld is None:
                ld = cand
            elif cand != ld:
                return False
    return ld is None or ld >= c


def to_ndhwc(t):
    """Plumbing: make an arbitrary [N,C,D,H,W] tensor dense NDHWC (no copy when it already is)."""
    if is_ndhwc(t):
        return t
    return t.contiguous(memory_format=torch.channels_last_3d)


class Split:
    """torch.cat((a, b), dim=1) held as two PLANES of one buffer (ru3d_tensor's split layout): `t` is the NDHWC tensor
    [2N, C/2, D, H, W] whose first N samples are `a` and last N samples `b`.  Quacks like the [N, C, D, H, W] concat for
    the wrappers in _ops (shape / dtype / device) and describes itself to the C ABI with cseg = C/2."""

    def __init__(self, t):
        n2, c, d, h, w = t.shape
        if n2 % 2 or not is_ndhwc(t) or not t.is_contiguous(memory_format=torch.channels_last_3d):
            raise Ru3dError("ru3d: a split concat must be a dense NDHWC tensor [2N, C/2, D, H, W]")
        self.t = t
        self.shape = torch.Size((n2 // 2, 2 * c, d, h, w))
        self.dtype = t.dtype
        self.device = t.device
        self.is_cuda = t.is_cuda

    def element_size(self):
        return self.t.element_size()

    def desc(self):
        n, c, d, h, w = self.shape
        _tls.device = self.t.device
        return Tensor(self.t.data_ptr(), n, d, h, w, c, c // 2, c // 2, n * d * h * w * (c // 2))


_GEOM = {}


def desc(t):
    """ru3d_tensor descriptor of an NDHWC-strided [N,C,D,H,W] torch tensor.  The (shape, strides) -> geometry
    part is memoised: the same few dozen layouts recur every step."""
    if isinstance(t, Split):
        return t.desc()
    if not t.is_cuda:
        require_device(t)
    _tls.device = t.device
    key = (t.shape, t.stride())
    g = _GEOM.get(key)
    if g is None:
        if not is_ndhwc(t):
            raise Ru3dError("ru3d: tensor of shape %s / strides %s is not NDHWC" % (tuple(t.shape), t.stride()))
        n, c, d, h, w = t.shape
        ld = c
        for size, stride, inner in ((w, t.stride(4), 1), (h, t.stride(3), w), (d, t.stride(2), h * w),
                                    (n, t.stride(0), d * h * w)):
            if size > 1:
                ld = stride // inner
                break
        g = (n, d, h, w, c, ld)
        if len(_GEOM) > 4096:
            _GEOM.clear()
        _GEOM[key] = g
    return Tensor(t.data_ptr(), *g)


def ref(d):
    return ctypes.byref(d) if d is not None else None


def ptr(t):
    if t is None:
        return None
    if t.is_cuda:
        dev = getattr(_tls, "device", None)
        if dev is not None and t.device != dev:
            raise Ru3dError("ru3d: operand on %s in a call assembled for %s (operands of one entry point must share a "
                            "device; note_device() / stream(device) name it)" % (t.device, dev))
    return ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------- workspace
_WS = {}


def workspace(nbytes, device, slot=0):
    """One growing scratch buffer per (device, stream[, slot]): every kernel that uses it runs in-order on that stream.
    slot > 0: a buffer of its own for scratch that must outlive the next kernels (deferred weight-gradient slabs)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream) if slot == 0 else \
        (device, torch.cuda.current_stream(device).cuda_stream, slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf
