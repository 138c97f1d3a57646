"""Drop-in `loss` module: Dice / DiceLoss / FocalLoss / HybirdLoss on the fused HIP loss kernels.

Import surface of the reference `loss.py` (functions `logits`, `flatten_and_tranpose_C`, `dice`,
`focal_loss`; classes `Dice`, `DiceLoss`, `FocalLoss`, `HybirdLoss`) plus the legacy aliases the
older training scripts import (`DiceCoef`, `FocalDiceCoefLoss`: reference nb_train_KITS19.py:4,
run_train.py:2 - no implementation of either name survives in the reference, so their semantics
here are this build's choice: `DiceCoef(weight=w) == Dice(weight_v=w)`,
`FocalDiceCoefLoss(d_weight=w) == HybirdLoss(weight_v=w)`).

Each module call is ONE fused forward pass over (logits, labels) on the device plus an on-device
finalize - no one-hot tensor, no per-class Python loop, no host synchronisation (the reference
syncs C+2 times per call, loss.py:231,241,246).  The returned value is a 0-dim tensor on the
logits' device that supports .backward(), .item() and torch.isnan like the reference's.

Bug-compatible behaviour kept (reference loss.py:68-69, 154-155, 236-237): `weight_c` and the
class-presence mask are accepted and have no effect; weights are `weight_v / sum|weight_v|`.
Labels outside [0, C) raise `RuntimeError("Class values must be smaller than num_classes.")` like the
reference's F.one_hot (loss.py:27) - without a host synchronisation of their own: the forward kernel
counts them into its state block, those 4 bytes ride to pinned host memory behind the kernel, and
`raise_on_bad_labels()` - called by the next loss call and by `Trainer` right after its own read-back of
the loss scalars - raises as soon as that copy has landed (the loss value of such a step is NaN).
`check_labels=True` / RU3D_CHECK_LABELS=1 checks before the launch instead (one sync per call).  With
C == 1 the reference only works for all-zero targets and so does this.
"""
import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

import _native as N
from _native import check, ptr, stream

_CHECK_LABELS = os.environ.get("RU3D_CHECK_LABELS", "0") == "1"


# --------------------------------------------------------------------------- functional helpers (API parity)
def logits(input):
    """(N, C, d1, ..., dn) -> class probabilities: softmax over C, sigmoid when C == 1."""
    return torch.softmax(input, dim=1) if input.size(1) > 1 else torch.sigmoid(input)


def flatten_and_tranpose_C(input, target):
    """(N, C, d1..dn), (N, d1..dn) -> (N*d1*..*dn, C) scores and int64 one-hot."""
    c = input.size(1)
    flat = input.reshape(input.size(0), c, -1).transpose(1, 2).reshape(-1, c)
    return flat, F.one_hot(target, num_classes=c).reshape(-1, c)


def dice(input, target, alpha=0.5, beta=0.5, smooth=1e-7):
    """Tversky index of two same-shaped tensors (probabilities or masks).  Device tensors go through
    the HIP reduction kernel; host tensors (trainer.evaluate_case works on numpy-derived CPU masks)
    are reduced by torch on the host - that evaluation path is not part of the training hot path."""
    if input.is_cuda:
        p = input.detach().reshape(-1).float().contiguous()
        g = target.detach().reshape(-1).float().contiguous()
        out = torch.empty((), dtype=torch.float32, device=p.device)
        N.note_device(p.device)
        ws = N.workspace(1024 * 3 * 8, p.device)
        check(N.lib.ru3d_tversky(ptr(p), ptr(g), p.numel(), alpha, beta, smooth, ptr(out), ptr(ws), ws.numel(),
                                 stream()), "tversky")
        return out
    p = input.reshape(-1)
    g = target.reshape(-1)
    tp = (p * g).sum()
    fn = ((1 - p) * g).sum()
    fp = (p * (1 - g)).sum()
    return (tp + smooth) / (tp + alpha * fn + beta * fp + smooth)


# --------------------------------------------------------------------------- deferred label check
_FLAG_RING = 64
_flag_host = {}                       # device -> [pinned int32 ring, next slot, events per slot]
_pending = collections.deque()        # (event, pinned ring, slot) in launch order
_BAD_LABELS = "Class values must be smaller than num_classes."      # F.one_hot's message (reference loss.py:27)


# graph.GraphedTrainStep sets this to a list while it captures a step: the state blocks of the captured loss calls, whose
# counts it then reads back after every replay (a copy queued OUTSIDE the graph, into the same pinned ring)
CAPTURE_SINK = [None]


def _note_label_flag(state):
    """Queue the D2H copy of the forward kernel's out-of-range label count (4 bytes, stream-ordered, no sync)."""
    dev = state.device
    if torch.cuda.is_current_stream_capturing():
        if CAPTURE_SINK[0] is not None:
            CAPTURE_SINK[0].append(state)
        return
    ent = _flag_host.get(dev)
    if ent is None:
        ent = _flag_host[dev] = [torch.zeros(_FLAG_RING, dtype=torch.int32).pin_memory(), 0, [None] * _FLAG_RING]
    ring, slot, events = ent
    if events[slot] is not None:
        events[slot].synchronize()      # 64 loss calls behind: long done
        raise_on_bad_labels()
    off = _BAD_OFF[0]
    ring[slot:slot + 1].copy_(state[off:off + 4].view(torch.int32), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    events[slot] = ev
    _pending.append((ev, ring, slot, events))
    ent[1] = (slot + 1) % _FLAG_RING


def raise_on_bad_labels(wait=False):
    """Raise F.one_hot's error for every finished loss call that saw a label outside [0, C).  Never blocks unless
    `wait`: call it right after a read-back of the loss (`.item()`, `.cpu()`) to learn about that very step."""
    while _pending:
        ev, ring, slot, events = _pending[0]
        if wait:
            ev.synchronize()
        elif not ev.query():
            return
        _pending.popleft()
        if events[slot] is ev:
            events[slot] = None
        if int(ring[slot]) > 0:
            _pending.clear()
            raise RuntimeError(_BAD_LABELS)


_BAD_OFF = [int(N.lib.ru3d_loss_state_bad_labels_offset())]


# --------------------------------------------------------------------------- fused loss
def _flat_strides(x):
    """(stride_n, stride_c, stride_v) of an (N, C, *spatial) tensor whose spatial dims collapse to one
    axis with a single stride (true for NCDHW-contiguous and for NDHWC tensors); None otherwise."""
    sizes, strides = x.shape[2:], x.stride()[2:]
    sv = None
    expect = None
    for size, stride in zip(reversed(sizes), reversed(strides)):
        if size == 1:
            continue
        if sv is None:
            sv, expect = stride, stride * size
        else:
            if stride != expect:
                return None
            expect = stride * size
    if sv is None:
        sv = 1
    return x.stride(0), x.stride(1), sv


class _FusedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, kind, gamma, weight_v, alpha, beta, smooth, check_labels):
        N.require_device(input, "loss input")
        if target.device != input.device:
            raise N.Ru3dError("loss: target is on %s but input on %s" % (target.device, input.device))
        if input.dim() < 2:
            raise N.Ru3dError("loss: input must be (N, C, d1, ..., dn)")
        n, c = input.shape[0], input.shape[1]
        if c > N.MAX_CLASSES:
            raise N.Ru3dError("loss: %d classes (max %d)" % (c, N.MAX_CLASSES))
        if tuple(target.shape) != (n,) + tuple(input.shape[2:]):
            raise N.Ru3dError("loss: target shape %s does not match input %s" % (tuple(target.shape),
                                                                                 tuple(input.shape)))
        x = input.detach()
        if x.dtype != torch.float32:
            x = x.float()
        st = _flat_strides(x)
        if st is None:
            x = x.contiguous()
            st = _flat_strides(x)
        v = 1
        for s in x.shape[2:]:
            v *= s
        if target.dtype == torch.int64:
            lab, lab_code = target.contiguous(), N.LABEL_I64
        elif target.dtype == torch.uint8:
            lab, lab_code = target.contiguous(), N.LABEL_U8
        else:
            lab, lab_code = target.long().contiguous(), N.LABEL_I64
        if check_labels or c == 1:
            # reference: F.one_hot(target, C) raises for labels >= C (always hit by C == 1 with labels {0,1})
            if lab.numel() and (int(lab.max()) >= c or int(lab.min()) < 0):
                raise RuntimeError(_BAD_LABELS)
        raise_on_bad_labels()          # an earlier call's verdict, if it has landed (no wait)
        dev = x.device
        state = torch.empty(N.lib.ru3d_loss_state_bytes(c), dtype=torch.uint8, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        ws = N.workspace(N.lib.ru3d_loss_workspace_bytes(n, v, c), dev)
        wv = None
        if weight_v is not None:
            if len(weight_v) != c:
                raise RuntimeError("weight_v has %d entries for %d classes" % (len(weight_v), c))
            wv = (N.ctypes.c_float * c)(*[float(w) for w in weight_v])
        check(N.lib.ru3d_loss_fwd(ptr(x), st[0], st[1], st[2], ptr(lab), lab_code, n, v, c, kind, float(gamma),
                                  N.ctypes.cast(wv, N.ctypes.c_void_p) if wv is not None else None, float(alpha),
                                  float(beta), float(smooth), ptr(state), ptr(out), ptr(ws), ws.numel(), stream()),
              "loss_fwd")
        if not (check_labels or c == 1):
            _note_label_flag(state)
        ctx.save_for_backward(x, lab, state)
        ctx.meta = (st, lab_code, n, v, c, float(gamma), input.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, lab, state = ctx.saved_tensors
        st, lab_code, n, v, c, gamma, in_dtype = ctx.meta
        g = gout.detach()
        if g.dtype != torch.float32 or g.device != x.device:
            g = g.to(device=x.device, dtype=torch.float32)
        g = g.reshape(1).contiguous()
        N.note_device(x.device)
        dz = torch.empty_like(x)   # preserve_format: same (dense) strides as the logits
        if dz.stride() != x.stride():
            dz = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)
        check(N.lib.ru3d_loss_bwd(ptr(x), st[0], st[1], st[2], ptr(lab), lab_code, n, v, c, gamma, ptr(state),
                                  ptr(g), ptr(dz), N.F32, stream()), "loss_bwd")
        if in_dtype != torch.float32:
            dz = dz.to(in_dtype)
        return dz, None, None, None, None, None, None, None, None


class _FusedLoss(nn.Module):
    _kind = None

    def _call(self, input, target, gamma, weight_v, alpha, beta, smooth):
        return _FusedLossFn.apply(input, target, self._kind, gamma, weight_v, alpha, beta, smooth,
                                  getattr(self, "check_labels", _CHECK_LABELS))


class Dice(_FusedLoss):
    """Weighted mean Tversky/Dice *score* (a metric: higher is better)."""
    _kind = N.LOSS_DICE

    def __init__(self, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
        super().__init__()
        self.weight_v = weight_v
        self.alpha = alpha
        self.beta = beta
        self.smooth = smooth

    def forward(self, input, target):
        return self._call(input, target, 2.0, self.weight_v, self.alpha, self.beta, self.smooth)


class DiceLoss(_FusedLoss):
    """sum_c w_c (1 - dice_c)."""
    _kind = N.LOSS_DICELOSS

    def __init__(self, weight_c=None, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
        super().__init__()
        self.weight_c = weight_c      # accepted, no effect (reference behaviour)
        self.weight_v = weight_v
        self.alpha = alpha
        self.beta = beta
        self.smooth = smooth

    def forward(self, input, target):
        return self._call(input, target, 2.0, self.weight_v, self.alpha, self.beta, self.smooth)


class FocalLoss(_FusedLoss):
    """sum_c w_c * C * mean_v(-(1 - p_c)^gamma * onehot_c * log p_c)."""
    _kind = N.LOSS_FOCAL

    def __init__(self, gamma=2, weight_c=None, weight_v=None):
        super().__init__()
        self.gamma = gamma
        self.weight_c = weight_c      # accepted, no effect (reference behaviour)
        self.weight_v = weight_v

    def forward(self, input, target):
        return self._call(input, target, self.gamma, self.weight_v, 0.5, 0.5, 1e-7)


class HybirdLoss(_FusedLoss):
    """sum_c w_c (1 - dice_c + focal_c): the training loss of the nb_train_* scripts."""
    _kind = N.LOSS_HYBIRD

    def __init__(self, gamma=2, weight_c=None, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
        super().__init__()
        self.weight_c = weight_c      # accepted, no effect (reference behaviour)
        self.weight_v = weight_v
        self.alpha = alpha
        self.beta = beta
        self.smooth = smooth
        self.gamma = gamma

    def forward(self, input, target):
        return self._call(input, target, self.gamma, self.weight_v, self.alpha, self.beta, self.smooth)


def focal_loss(input, target, gamma=2, weight_c=None, weight_v=None):
    """Functional focal loss on flattened (N*V, C) scores and one-hot targets (reference signature)."""
    labels = target.argmax(dim=1) if target.dim() == 2 else target
    x = input.t().unsqueeze(0)          # (1, C, N*V): class axis second, voxels last
    return _FusedLossFn.apply(x, labels.reshape(1, -1), N.LOSS_FOCAL, gamma, weight_v, 0.5, 0.5, 1e-7,
                              _CHECK_LABELS)


# --------------------------------------------------------------------------- legacy names
class DiceCoef(Dice):
    """Pre-refactor name used by nb_train_KITS19.py / run_train.py: DiceCoef(weight=[...])."""

    def __init__(self, weight=None, alpha=0.5, beta=0.5, smooth=1e-7):
        super().__init__(weight_v=weight, alpha=alpha, beta=beta, smooth=smooth)


class FocalDiceCoefLoss(HybirdLoss):
    """Pre-refactor name used by nb_train_KITS19.py / run_train.py: FocalDiceCoefLoss(d_weight=[...])."""

    def __init__(self, d_weight=None, gamma=2, alpha=0.5, beta=0.5, smooth=1e-7):
        super().__init__(gamma=gamma, weight_v=d_weight, alpha=alpha, beta=beta, smooth=smooth)
