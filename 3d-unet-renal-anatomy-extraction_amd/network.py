"""Drop-in `network` module: the reference's U-Net family on the MI355X-native hot path.

Same import surface as the reference `network.py` (class names, constructor signatures, attribute
names, `state_dict` keys and parameter creation order - so `torch.manual_seed(s); ResUnet3D(...)`
yields the reference's initial weights and a reference checkpoint loads with strict=True).
What differs is what runs: the blocks `ResUnet3D` is assembled from (ResBlock, ResBlockStack,
ConvTrans3D, UpConcat, stem/head convs) execute as hand-written HIP kernels from libru3d.so through
`_ops` (NDHWC activations, fp32 or bf16 storage, fp32 accumulation).  nn.Conv3d / nn.ConvTranspose3d
sub-modules are kept as *parameter containers only*; their torch forward is never called on the
native path.

Reference behaviour implemented (file:line in the reference repo):
  ResUnet3D 104-132, generate_paired_features 135-141, ConvTrans3D 298-320, UpConcat 323-350,
  ResBlock 374-416, ResBlockStack 419-449, Unet 470-565.
Out-of-hot-path variants (ConvBlock*, RecBlock, ResRecBlock, AttBlock, MaxPoolBlock, BatchNorm /
attention configurations: reference 153-295, 353-371, 452-463) are ordinary torch modules.

Precision: `set_compute_dtype(model, torch.bfloat16)` (or env RU3D_DTYPE=bf16) selects bf16 storage
with fp32 accumulation, torch.float16 the reference's apex-O1 arithmetic (fp16 storage, fp32 accumulation; train it
with optim.LossScaler / Trainer.fit(use_amp=True)); the default float32 is the parity mode.  Logits are always fp32.
"""
import os

import torch
import torch.nn as nn

import _native as N
import _ops as ops

_DEFAULT_DTYPE = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                  "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16,
                  "half": torch.float16}[os.environ.get("RU3D_DTYPE", "fp32").lower()]


_STORAGE_DTYPES = (torch.float32, torch.bfloat16, torch.float16)


def set_compute_dtype(model, dtype):
    """Select the activation/weight storage dtype of every native U-Net inside `model`."""
    if dtype not in _STORAGE_DTYPES:
        raise ValueError("compute dtype must be one of %s" % (_STORAGE_DTYPES,))
    for m in model.modules():
        if isinstance(m, Unet):
            m.compute_dtype = dtype
            m._configure_native()
    return model


def set_checkpointing(model, enabled=True):
    """Activation checkpointing of the native ResBlocks (BASELINE config 5): keep only each block's input, output
    and InstanceNorm statistics for backward and recompute its three interior tensors there.  Gradients are
    bit-identical to the un-checkpointed run; saved activations drop to roughly two fifths."""
    for m in model.modules():
        if isinstance(m, ResBlock):
            m._checkpoint = bool(enabled)
    return model


# --------------------------------------------------------------------------- feature plans
def generate_paired_features(num_pool, num_features):
    widths = [num_features * (1 << i) for i in range(num_pool + 1)]
    plan = [[c, c] for c in widths]                       # down path + bottom
    plan += [[c, c] for c in reversed(widths[:-1])]       # up path
    return plan


def generate_paired_features2(num_pool, num_features):
    widths = [num_features * (1 << i) for i in range(num_pool + 1)]
    plan = [[widths[i], widths[i + 1]] for i in range(num_pool)]
    plan.append([widths[num_pool], widths[num_pool]])
    plan += [[c, c] for c in reversed(widths[:-1])]
    return plan


def none_fn(level):
    return {}


# --------------------------------------------------------------------------- native eligibility
def _is_plain_conv3(conv):
    return (type(conv) is nn.Conv3d and conv.kernel_size == (3, 3, 3) and conv.padding == (1, 1, 1)
            and conv.dilation == (1, 1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.bias is not None)


def _is_plain_conv1(conv):
    return (type(conv) is nn.Conv3d and conv.kernel_size == (1, 1, 1) and conv.padding == (0, 0, 0)
            and conv.dilation == (1, 1, 1) and conv.groups == 1 and conv.stride in ((1, 1, 1), (2, 2, 2))
            and conv.bias is not None)


def _is_plain_in(norm):
    return (type(norm) is nn.InstanceNorm3d and not norm.affine and not norm.track_running_stats
            and abs(norm.eps - ops.IN_EPS) < 1e-12)


def _is_plain_bn(norm):
    return (type(norm) is nn.BatchNorm3d and norm.affine and norm.track_running_stats)


def _bn_eval_stats(norm, n, cpad):
    """Inference-mode BatchNorm3d (reference blocks built with norm_op=nn.BatchNorm3d, network.py:38-69) is the
    per-channel affine map gamma * (y - running_mean) / sqrt(running_var + eps) + beta.  Written for the InstanceNorm
    apply kernel (out = lrelu((y - mean) * scale + res)): scale = gamma / sqrt(var + eps), mean = running_mean -
    beta / scale, the same for every sample; pad lanes (channel-padded nets) get mean 0 / scale 1 so they stay 0."""
    scale = norm.weight.detach().float() / torch.sqrt(norm.running_var.float() + norm.eps)
    safe = torch.where(scale.abs() < 1e-20, torch.full_like(scale, 1e-20), scale)
    mean = norm.running_mean.float() - norm.bias.detach().float() / safe
    c = scale.numel()
    if cpad > c:
        mean = torch.nn.functional.pad(mean, (0, cpad - c), value=0.0)
        safe = torch.nn.functional.pad(safe, (0, cpad - c), value=1.0)
    return mean.repeat(n).contiguous(), safe.repeat(n).contiguous()


def _inference_mode(module):
    return (not module.training) and (not torch.is_grad_enabled())


def _bn_on_device(module):
    """A BatchNorm block runs on the native kernels: in inference mode (running statistics) and in training mode."""
    return _inference_mode(module) or (module.training and _bn_native_train())


def _bn_native_train():
    """BatchNorm blocks train on the native kernels (ops.ResBlockBNFn / UpBNFn); RU3D_BN_TRAIN=torch keeps them torch
    modules in training mode (the round-2 behaviour, kept as a cross-check)."""
    return os.environ.get("RU3D_BN_TRAIN", "native") != "torch"


def _is_plain_lrelu(act):
    return type(act) is nn.LeakyReLU and abs(act.negative_slope - ops.LRELU_SLOPE) < 1e-12


# --------------------------------------------------------------------------- generic (torch) blocks
class ConvBlock(nn.Module):
    """conv -> dropout -> norm -> nonlin (out of the native hot path; plain torch ops)."""

    def __init__(self, in_channels, out_channels, conv_op=nn.Conv3d,
                 conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.conv = conv_op(in_channels, out_channels, **conv_kwargs)
        self.dropout = dropout_op(**dropout_kwargs) if dropout_op else None
        self.norm = norm_op(out_channels, **norm_kwargs)
        self.nonlin = nonlin_op(**nonlin_kwargs)

    def forward(self, x):
        x = self.conv(x)
        if self.dropout:
            x = self.dropout(x)
        return self.nonlin(self.norm(x))


class ConvBlockStack(nn.Module):
    def __init__(self, in_channels, out_channels, num_stacks=2, conv_op=nn.Conv3d,
                 conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        common = dict(conv_op=conv_op, conv_kwargs=conv_kwargs, dropout_op=dropout_op,
                      dropout_kwargs=dropout_kwargs, norm_op=norm_op, norm_kwargs=norm_kwargs,
                      nonlin_op=nonlin_op, nonlin_kwargs=nonlin_kwargs)
        self.conv_blocks = nn.ModuleList(
            ConvBlock(in_channels if i == 0 else out_channels, out_channels, **common) for i in range(num_stacks))

    def forward(self, x):
        for blk in self.conv_blocks:
            x = blk(x)
        return x


class RecBlock(nn.Module):
    """Recurrent conv block (R2U-Net style), torch ops."""

    def __init__(self, out_channels, t=2, conv_op=nn.Conv3d, conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.t = t
        self.out_channels = out_channels
        self.conv = ConvBlock(out_channels, out_channels, conv_op=conv_op, conv_kwargs=conv_kwargs,
                              dropout_op=dropout_op, dropout_kwargs=dropout_kwargs, norm_op=norm_op,
                              norm_kwargs=norm_kwargs, nonlin_op=nonlin_op, nonlin_kwargs=nonlin_kwargs)

    def forward(self, x):
        out = self.conv(x)
        for _ in range(self.t):
            out = self.conv(out + x)
        return out


class ResRecBlock(nn.Module):
    """Residual recurrent block, torch ops."""

    def __init__(self, in_channels, out_channels, t=2, conv_op=nn.Conv3d,
                 conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.t = t
        self.in_channels = in_channels
        self.out_channels = out_channels
        common = dict(conv_op=conv_op, conv_kwargs=conv_kwargs, dropout_op=dropout_op,
                      dropout_kwargs=dropout_kwargs, norm_op=norm_op, norm_kwargs=norm_kwargs,
                      nonlin_op=nonlin_op, nonlin_kwargs=nonlin_kwargs)
        self.rcnn = nn.Sequential(RecBlock(out_channels, t=t, **common), RecBlock(out_channels, t=t, **common))
        self.conv = conv_op(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        skip = self.conv(x) if self.in_channels != self.out_channels else x
        return self.rcnn(skip) + skip


class AttBlock(nn.Module):
    """Attention gate sharing one 1x1 conv (reference network.py:353-371).  Inside a native UpConcat it runs as
    ops.AttGateFn (conv kernels + ru3d_pointwise); called on its own it is the plain torch composition."""

    def __init__(self, out_channels, conv_op=nn.Conv3d, nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.conv = conv_op(out_channels, out_channels, kernel_size=1)
        self.lrelu = nonlin_op(**nonlin_kwargs)
        self.active = nn.Sigmoid()
        self._native = _is_plain_conv1(self.conv) and self.conv.stride == (1, 1, 1) and _is_plain_lrelu(self.lrelu)

    def forward(self, x, gate):
        x = self.conv(x)
        g = self.conv(gate)
        rate = self.active(self.conv(self.lrelu(x + g)))
        return x * rate


class MaxPoolBlock(nn.Module):
    def __init__(self, in_channels, out_channels, pool_op=nn.MaxPool3d, pool_kwargs={'kernel_size': 2, 'stride': 2}):
        super().__init__()
        self.pool = pool_op(**pool_kwargs)

    def forward(self, x):
        return self.pool(x)


# --------------------------------------------------------------------------- native blocks
class ConvTrans3D(nn.Module):
    """ConvTranspose3d(k3,s2,p1) -> zero pad on the far side of each axis -> norm -> nonlin."""

    def __init__(self, in_channels, out_channels, norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.up = nn.Sequential(
            nn.ConvTranspose3d(in_channels, out_channels, kernel_size=3, stride=2, padding=1),
            nn.ConstantPad3d(padding=(0, 1, 0, 1, 0, 1), value=0),
            norm_op(out_channels, **norm_kwargs),
            nonlin_op(**nonlin_kwargs))
        self._native = _is_plain_in(self.up[2]) and _is_plain_lrelu(self.up[3])
        self._bn_eval = _is_plain_bn(self.up[2]) and _is_plain_lrelu(self.up[3])   # native in inference mode only
        self._pad = False          # set by Unet: activations carry channels zero-padded to multiples of 32

    def _pack_specs(self):
        """The packed forms ops.UpFn asks for in a training pass."""
        wt, bt = self.up[0].weight, self.up[0].bias
        cin, cout = wt.shape[0], wt.shape[1]
        cin_seg = cin if self._pad else 0
        cout_seg = cout if self._pad else 0
        specs = [(wt, N.ROLE_CONVT_FWD, 2, cout_seg, cin_seg), (wt, N.ROLE_CONVT_DGRAD, 2, cout_seg, cin_seg)]
        if cout_seg and bt is not None:
            specs.append((bt, N.ROLE_BIAS, 1, cout_seg, 0))
        return specs

    def forward(self, x, skip=None, link=None):
        """`skip` is an extension used by UpConcat: returns cat((up(x), skip), dim=1) written in place.
        `link`: ops.SkipLink of this skip connection (Unet passes it; see _ops.SkipLink)."""
        if self._native and x.is_cuda:
            return ops.UpFn.apply(x, self.up[0].weight, self.up[0].bias, skip, self._pad, link)
        if self._native:
            N.require_device(x, "ConvTrans3D input")
        if self._bn_eval and x.is_cuda and _inference_mode(self):
            u = self._forward_bn_eval(x)
            return u if skip is None else ops.concat_channels(u, skip)
        if self._bn_eval and x.is_cuda and self.training and _bn_native_train():
            bn = self.up[2]
            return ops.UpBNFn.apply(x, self.up[0].weight, self.up[0].bias, bn.weight, bn.bias, skip, self._pad, bn)
        y = self.up(x)
        return y if skip is None else torch.cat((y, skip), dim=1)

    def _forward_bn_eval(self, x):
        """ConvTranspose3d + far zero pad + inference BatchNorm + LeakyReLU on the native kernels (no tape)."""
        wt, bt = self.up[0].weight, self.up[0].bias
        x = ops.as_grad(x, x.dtype)
        cin, cout = wt.shape[0], wt.shape[1]
        cin_seg = ops.seg_of(cin, x.shape[1]) if self._pad else 0
        cout_seg = cout if self._pad else 0
        cp = ops.padded_dim(cout, cout_seg)
        specs = [(wt, N.ROLE_CONVT_FWD, 2, cout_seg, cin_seg)]
        if cout_seg and bt is not None:
            specs.append((bt, N.ROLE_BIAS, 1, cout_seg, 0))
        packs = ops.pack_weights(specs, x.dtype)
        b = ops._f32_view(packs[-1], cp) if (cout_seg and bt is not None) else bt
        y = ops.convt_fwd(x, packs[0], b, cp)
        mean, scale = _bn_eval_stats(self.up[2], x.shape[0], cp)
        return ops.in_lrelu_fwd(y, mean, scale)


class UpConcat(nn.Module):
    def __init__(self, in_channels, out_channels, conv_trans_op=ConvTrans3D, attention=False,
                 att_conv_op=nn.Conv3d, norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.attention = attention
        self.conv_trans = conv_trans_op(in_channels, out_channels, norm_op=norm_op, norm_kwargs=norm_kwargs,
                                        nonlin_op=nonlin_op, nonlin_kwargs=nonlin_kwargs)
        if attention:
            self.att_gate = AttBlock(out_channels, conv_op=att_conv_op, nonlin_op=nonlin_op,
                                     nonlin_kwargs=nonlin_kwargs)

    def forward(self, x, skip, link=None):
        if not self.attention and isinstance(self.conv_trans, ConvTrans3D):
            return self.conv_trans(x, skip, link)    # up-sampled channels first, skip second
        if (self.attention and isinstance(self.conv_trans, ConvTrans3D) and self.att_gate._native and x.is_cuda
                and (self.conv_trans._native or (self.conv_trans._bn_eval and _bn_on_device(self)))):
            up = self.conv_trans(x)
            return ops.AttGateFn.apply(skip, up, self.att_gate.conv.weight, self.att_gate.conv.bias,
                                       self.conv_trans._pad)
        x = self.conv_trans(x)
        if self.attention:
            skip = self.att_gate(skip, x)
        return torch.cat((x, skip), dim=1)


class ResBlock(nn.Module):
    """conv1(stride) -> dropout -> norm -> nonlin -> conv2 -> norm -> (+ skip) -> nonlin."""

    def __init__(self, in_channels, out_channels, stride=1, conv_op=nn.Conv3d,
                 conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.stride = stride
        self.conv1 = conv_op(in_channels, out_channels, stride=stride, **conv_kwargs)
        self.conv2 = conv_op(out_channels, out_channels, **conv_kwargs)
        self.dropout = dropout_op(**dropout_kwargs) if dropout_op else None
        self.norm = norm_op(out_channels, **norm_kwargs)
        self.nonlin = nonlin_op(**nonlin_kwargs)
        # always constructed (so it is always in the state_dict), used only when the shapes differ
        self.skip_conv = conv_op(in_channels, out_channels, kernel_size=1, stride=stride)
        self._native = (_is_plain_conv3(self.conv1) and _is_plain_conv3(self.conv2)
                        and _is_plain_conv1(self.skip_conv) and _is_plain_in(self.norm)
                        and _is_plain_lrelu(self.nonlin)
                        and (self.dropout is None or type(self.dropout) is nn.Dropout3d))
        self._bn_eval = (_is_plain_conv3(self.conv1) and _is_plain_conv3(self.conv2) and _is_plain_conv1(self.skip_conv)
                         and _is_plain_bn(self.norm) and _is_plain_lrelu(self.nonlin))   # native in inference mode only
        self._forced_keep = None   # tests: inject a recorded [N, C] keep mask instead of drawing one
        self._pad = False          # set by Unet: activations carry channels zero-padded to multiples of 32
        self._in_segs = 1          # set by Unet: 2 when the input is the padded concat [up | skip]
        self._checkpoint = False   # set_checkpointing(): recompute the block's interior in backward

    @property
    def uses_skip_conv(self):
        return self.in_channels != self.out_channels or self.stride != 1

    def _drop_scale(self, x, planar=False):
        if self.dropout is None or not self.training:
            return None
        p = float(self.dropout.p)
        n = x.shape[0] // 2 if planar else x.shape[0]
        c = ops.cpad(self.out_channels) if self._pad else self.out_channels
        if self._forced_keep is not None:
            keep = self._forced_keep.to(device=x.device, dtype=torch.float32).reshape(n, -1)
            if keep.shape[1] != c:      # pad lanes hold exact zeros whatever their factor is
                keep = torch.nn.functional.pad(keep, (0, c - keep.shape[1]), value=1.0)
            return (keep.reshape(-1) / (1.0 - p)).contiguous()
        if p == 0.0:
            return None
        pooled = ops.take_dropout(n, c, p)      # Unet.forward drew the whole pass's factors in one launch
        return pooled if pooled is not None else ops.dropout_scale(n, c, p, x.device)

    def _drop_spec(self, n):
        """(n, c, p) of the draw this block will make in training mode, or None (same conditions as _drop_scale)."""
        if (not self._native or self.dropout is None or not self.training or self._forced_keep is not None
                or float(self.dropout.p) == 0.0):
            return None
        return (n, ops.cpad(self.out_channels) if self._pad else self.out_channels, float(self.dropout.p))

    def _pack_specs(self):
        """The packed weight forms ops.ResBlockFn asks for in a training pass (same tuples: ops.prepack keys on them)."""
        w1, w2 = self.conv1.weight, self.conv2.weight
        cout, cin = w1.shape[0], w1.shape[1]
        cout_seg = cout if self._pad else 0
        cin_seg = (cin // self._in_segs) if self._pad else 0
        specs = [(w1, N.ROLE_CONV_FWD, self.stride, cout_seg, cin_seg), (w1, N.ROLE_CONV_DGRAD, self.stride, cout_seg, cin_seg),
                 (w2, N.ROLE_CONV_FWD, 1, cout_seg, cout_seg), (w2, N.ROLE_CONV_DGRAD, 1, cout_seg, cout_seg)]
        biases = [self.conv1.bias, self.conv2.bias]
        if self.uses_skip_conv:
            ws = self.skip_conv.weight
            specs += [(ws, N.ROLE_CONV_FWD, self.stride, cout_seg, cin_seg), (ws, N.ROLE_CONV_DGRAD, self.stride, cout_seg, cin_seg)]
            biases.append(self.skip_conv.bias)
        if cout_seg:
            specs += [(b, N.ROLE_BIAS, 1, cout_seg, 0) for b in biases if b is not None]
        return specs

    def forward(self, x, in_link=None, out_link=None, planar=False):
        """in_link / out_link: ops.SkipLink objects Unet passes to the pooling block / the last encoder block of a
        level (see _ops.SkipLink); standalone use leaves them None.  planar: x is the split concat ops.UpFn made on the
        full-resolution level ([2N, C/2, D, H, W]: the two halves as planes of one buffer)."""
        if self._native and x.is_cuda:
            skip_w = self.skip_conv.weight if self.uses_skip_conv else None
            skip_b = self.skip_conv.bias if self.uses_skip_conv else None
            return ops.ResBlockFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                        skip_w, skip_b, self.stride, self._drop_scale(x, planar),
                                        self._in_segs if self._pad else 0,
                                        self._checkpoint and torch.is_grad_enabled(), in_link, out_link, planar)
        if self._native:
            N.require_device(x, "ResBlock input")
        if self._bn_eval and x.is_cuda and _inference_mode(self):
            return self._forward_bn_eval(x)
        if (self._bn_eval and x.is_cuda and self.training and _bn_native_train()
                and (self.dropout is None or type(self.dropout) is nn.Dropout3d)):
            skip_w = self.skip_conv.weight if self.uses_skip_conv else None
            skip_b = self.skip_conv.bias if self.uses_skip_conv else None
            return ops.ResBlockBNFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                          skip_w, skip_b, self.norm.weight, self.norm.bias, self.stride,
                                          self._drop_scale(x), self._in_segs if self._pad else 0, self.norm)
        skip = self.skip_conv(x) if self.uses_skip_conv else x
        x = self.conv1(x)
        if self.dropout:
            x = self.dropout(x)
        x = self.nonlin(self.norm(x))
        x = self.conv2(x)
        return self.nonlin(self.norm(x) + skip)


def _resblock_bn_eval(self, x):
    """ResBlock with BatchNorm in inference mode on the native kernels (no tape; Dropout3d is the identity in eval):
    conv1 -> BN -> lrelu -> conv2 -> BN -> (+ skip) -> lrelu with the ONE shared norm's running statistics."""
    x = ops.as_grad(x, x.dtype)
    sd = x.dtype
    cout, cin = self.conv1.weight.shape[0], self.conv1.weight.shape[1]
    cin_seg = ops.seg_of(cin, x.shape[1], self._in_segs) if self._pad else 0
    cout_seg = cout if self._pad else 0
    cp = ops.padded_dim(cout, cout_seg)
    specs = [(self.conv1.weight, N.ROLE_CONV_FWD, self.stride, cout_seg, cin_seg),
             (self.conv2.weight, N.ROLE_CONV_FWD, 1, cout_seg, cout_seg)]
    if self.uses_skip_conv:
        specs.append((self.skip_conv.weight, N.ROLE_CONV_FWD, self.stride, cout_seg, cin_seg))
    nw = len(specs)
    biases = [self.conv1.bias, self.conv2.bias, self.skip_conv.bias if self.uses_skip_conv else None]
    if cout_seg:
        specs += [(b, N.ROLE_BIAS, 1, cout_seg, 0) for b in biases if b is not None]
    packs = ops.pack_weights(specs, sd)
    if cout_seg:
        it = iter(packs[nw:])
        biases = [(ops._f32_view(next(it), cp) if b is not None else None) for b in biases]
    mean, scale = _bn_eval_stats(self.norm, x.shape[0], cp)
    y1 = ops.conv_fwd(x, packs[0], biases[0], cp, 3, self.stride)
    a1 = ops.in_lrelu_fwd(y1, mean, scale)
    y2 = ops.conv_fwd(a1, packs[1], biases[1], cp, 3, 1)
    skip = ops.conv_fwd(x, packs[2], biases[2], cp, 1, self.stride) if self.uses_skip_conv else x
    return ops.in_lrelu_fwd(y2, mean, scale, res=skip)


ResBlock._forward_bn_eval = _resblock_bn_eval


class ResBlockStack(nn.Module):
    def __init__(self, in_channels, out_channels, stride=1, num_stacks=2, conv_op=nn.Conv3d,
                 conv_kwargs={'kernel_size': 3, 'padding': 1},
                 dropout_op=nn.Dropout3d, dropout_kwargs={'p': 0.5, 'inplace': True},
                 norm_op=nn.InstanceNorm3d, norm_kwargs={},
                 nonlin_op=nn.LeakyReLU, nonlin_kwargs={'inplace': True}):
        super().__init__()
        common = dict(conv_op=conv_op, conv_kwargs=conv_kwargs, dropout_op=dropout_op,
                      dropout_kwargs=dropout_kwargs, norm_op=norm_op, norm_kwargs=norm_kwargs,
                      nonlin_op=nonlin_op, nonlin_kwargs=nonlin_kwargs)
        self.res_blocks = nn.ModuleList(
            ResBlock(in_channels if i == 0 else out_channels, out_channels, stride=stride if i == 0 else 1,
                     **common) for i in range(num_stacks))

    def forward(self, x, out_link=None):
        last = len(self.res_blocks) - 1
        for i, blk in enumerate(self.res_blocks):
            x = blk(x, out_link=out_link) if (i == last and out_link is not None) else blk(x)
        return x


class Unet(nn.Module):
    """Generic encoder/decoder assembler: stem conv, [encode_i, pool_i] x P, bottom encode,
    [up_i(x, skip_i), decode_i] x P (deepest first), 1x1x1 head.  Block classes are constructor
    arguments, exactly as in the reference."""

    def __init__(self, in_channels, out_channels, paired_features,
                 pool_block=MaxPoolBlock, pool_kwargs={}, pool_kwargs_fn=none_fn,
                 up_block=UpConcat, up_kwargs={}, up_kwargs_fn=none_fn,
                 encode_block=ConvBlockStack, encode_kwargs={}, encode_kwargs_fn=none_fn,
                 decode_block=ConvBlockStack, decode_kwargs={}, decode_kwargs_fn=none_fn,
                 conv_op=nn.Conv3d):
        super().__init__()
        num_pairs = len(paired_features)
        assert (num_pairs % 2) == 1, 'Number of paired features must be odd number.'
        self.num_pool = num_pairs // 2
        assert self.num_pool > 0, 'At least one pool.'
        pf = paired_features
        pools, ups, encs, decs = [], [], [], []
        # construction order (pool, up, encode, decode per level; then bottom, stem, head) is part of
        # the contract: it fixes the RNG stream of the default initialisation.
        for i in range(self.num_pool):
            mirror = num_pairs - i - 1
            pools.append(pool_block(pf[i][1], pf[i + 1][0], **pool_kwargs_fn(i), **pool_kwargs))
            ups.append(up_block(pf[mirror - 1][1], pf[mirror][0], **up_kwargs_fn(i), **up_kwargs))
            encs.append(encode_block(pf[i][0], pf[i][1], **encode_kwargs_fn(i), **encode_kwargs))
            decs.append(decode_block(pf[mirror][0] + pf[i][1], pf[mirror][1], **decode_kwargs_fn(i),
                                     **decode_kwargs))
        encs.append(encode_block(pf[self.num_pool][0], pf[self.num_pool][1], **encode_kwargs_fn(self.num_pool),
                                 **encode_kwargs))
        self.pool_blocks = nn.ModuleList(pools)
        self.up_blocks = nn.ModuleList(ups)
        self.encode_blocks = nn.ModuleList(encs)
        self.decode_blocks = nn.ModuleList(decs)
        self.conv = conv_op(in_channels, pf[0][0], kernel_size=3, padding=1)
        self.fc = conv_op(pf[num_pairs - 1][1], out_channels, kernel_size=1)
        self.compute_dtype = _DEFAULT_DTYPE
        self._native_io = _is_plain_conv3(self.conv) and _is_plain_conv1(self.fc)
        self._pad = False
        self._linked = False
        self._bn_blocks = None
        self._in_chain = False
        self._pad_bn = False
        self._configure_native()

    def _native_chain(self, bn_ok=False):
        """The native blocks between stem and head, or None when any block on the path is a torch module.
        bn_ok: also accept a chain of BatchNorm blocks (native in inference mode only)."""
        blocks = []
        for blk in list(self.pool_blocks) + list(self.encode_blocks) + list(self.decode_blocks):
            if isinstance(blk, ResBlockStack):
                blocks += list(blk.res_blocks)
            else:
                blocks.append(blk)
        kinds = set()
        for blk in blocks:
            if not (isinstance(blk, ResBlock) and (blk._native or blk._bn_eval)):
                return None
            kinds.add("in" if blk._native else "bn")
        for up in self.up_blocks:
            if not (isinstance(up, UpConcat) and isinstance(up.conv_trans, ConvTrans3D)
                    and (up.conv_trans._native or up.conv_trans._bn_eval)
                    and (not up.attention or up.att_gate._native)):
                return None
            kinds.add("in" if up.conv_trans._native else "bn")
            blocks.append(up.conv_trans)
        if len(kinds) != 1 or not self._native_io:
            return None
        if "bn" in kinds and not bn_ok:
            return None
        return blocks

    def _configure_native(self):
        """Channel padding: with a 16-bit storage dtype, widths that are not multiples of 32 (the reference's default
        num_features = 30 gives 30/60/120/240/480, network.py:107) run on the MFMA kernels with every activation
        between stem and head zero-padded to the next multiple of 32 (the pad lanes stay exact zeros through conv,
        InstanceNorm, LeakyReLU and their gradients; weights are packed with zero rows/columns and weight gradients
        are cut back to the parameters' shapes).  Forward hooks on inner blocks see the padded channel counts."""
        chain = self._native_chain()
        bn_chain = self._native_chain(bn_ok=True) if chain is None else None
        self._bn_blocks = bn_chain
        self._in_chain = chain is not None
        self._pad_bn = False
        if bn_chain is not None:
            wb = [c for blk in bn_chain for c in (blk.in_channels, blk.out_channels)]
            self._pad_bn = (self.compute_dtype != torch.float32 and any(c % 32 for c in wb)
                            and all(ops.cpad(c) * 2 <= 3 * c for c in wb)
                            and os.environ.get("RU3D_PAD_CHANNELS", "1") != "0")
        widths = []
        if chain is not None:
            for blk in chain:
                widths += [blk.in_channels, blk.out_channels]
        # worth it while the padding adds at most half again to every width (F >= 22); narrower toy models keep the
        # direct kernels rather than carry 2-4x the activation bytes
        pad = (chain is not None and self.compute_dtype != torch.float32 and any(c % 32 for c in widths)
               and all(ops.cpad(c) * 2 <= 3 * c for c in widths)
               and os.environ.get("RU3D_PAD_CHANNELS", "1") != "0")
        self._pad = pad
        # skip connections of an all-native net go through ops.SkipLink (no concat copy, no autograd add)
        self._linked = (chain is not None and not any(getattr(up, "attention", False) for up in self.up_blocks)
                        and all(isinstance(b, (ResBlock, ResBlockStack)) for b in self.encode_blocks)
                        and all(isinstance(b, ResBlock) and b.uses_skip_conv for b in self.pool_blocks)
                        and os.environ.get("RU3D_SKIP_LINK", "1") != "0")
        for blk in (chain or []):
            blk._pad = pad
        for blk in self.decode_blocks:      # their input is cat((up, skip)): two padded segments
            first = blk.res_blocks[0] if isinstance(blk, ResBlockStack) else blk
            if isinstance(first, ResBlock):
                first._in_segs = 2

    def _storage_dtype(self):
        """16-bit storage only where the blocks behind the stem consume it: an all-native InstanceNorm chain, or a
        BatchNorm chain in inference mode; torch-module blocks (BatchNorm training, custom blocks) get fp32."""
        if self._in_chain or (self._bn_blocks is not None and _bn_on_device(self)):
            return self.compute_dtype
        return torch.float32

    def _stem(self, x):
        if self._native_io and x.is_cuda:
            sd = self._storage_dtype()
            return ops.ConvFn.apply(x, self.conv.weight, self.conv.bias, 1, sd, sd, False, self._pad)
        if self._native_io:
            N.require_device(x, "Unet input")
        return self.conv(x)

    def _head(self, x):
        if self._native_io and x.is_cuda:
            return ops.ConvFn.apply(x, self.fc.weight, self.fc.bias, 1, x.dtype, torch.float32, self._pad, False)
        return self.fc(x)

    def _prepack(self):
        """One packing pass for the whole model in front of a training step (ops.prepack): every block of the linked
        native chain, the stem and the head."""
        sd = self._storage_dtype()
        cs = self.conv.weight.shape[0] if self._pad else 0
        specs = [(self.conv.weight, N.ROLE_CONV_FWD, 1, cs, 0)]
        if cs and self.conv.bias is not None:
            specs.append((self.conv.bias, N.ROLE_BIAS, 1, cs, 0))
        for blk in self._native_chain() or []:
            specs += blk._pack_specs()
        ci = self.fc.weight.shape[1] if self._pad else 0
        specs += [(self.fc.weight, N.ROLE_CONV_FWD, 1, 0, ci), (self.fc.weight, N.ROLE_CONV_DGRAD, 1, 0, ci)]
        ops.prepack(specs, sd)

    def _set_bn_pad(self, pad):
        """BatchNorm nets are native in inference mode only, so their channel padding is decided per forward."""
        if pad != self._pad:
            self._pad = pad
            for blk in self._bn_blocks:
                blk._pad = pad

    def forward(self, x):
        if self._bn_blocks is not None:
            self._set_bn_pad(self._pad_bn and x.is_cuda and _bn_on_device(self))
        linked = self._linked and x.is_cuda
        if linked and torch.is_grad_enabled() and self._storage_dtype() != torch.float32:
            self._prepack()
        x = self._stem(x)
        skips, links = [], []
        if linked and self.training:
            # every Dropout3d factor of the pass from one launch (ops.prefill_dropout), in the blocks' execution order
            order = []
            for i in range(self.num_pool):
                order += [self.encode_blocks[i], self.pool_blocks[i]]
            order += [self.encode_blocks[-1]]
            for i in reversed(range(self.num_pool)):
                order += [self.decode_blocks[i]]
            specs = []
            for m in order:
                for blk in (m.res_blocks if isinstance(m, ResBlockStack) else [m]):
                    sp = blk._drop_spec(x.shape[0]) if isinstance(blk, ResBlock) else None
                    if sp is not None:
                        specs.append(sp)
            ops.prefill_dropout(specs, x.device)
        for i in range(self.num_pool):
            link = None
            if linked:
                cu = self.up_blocks[i].conv_trans.out_channels
                cd = self.decode_blocks[i].out_channels if isinstance(self.decode_blocks[i], ResBlock) else 0
                link = ops.SkipLink(ops.cpad(cu) if self._pad else cu, ops.cpad(cd) if self._pad else cd)
                x = self.encode_blocks[i](x, out_link=link)
            else:
                x = self.encode_blocks[i](x)
            skips.append(x)
            links.append(link)
            x = self.pool_blocks[i](x, in_link=link) if linked else self.pool_blocks[i](x)
        x = self.encode_blocks[-1](x)
        for i in reversed(range(self.num_pool)):
            x = self.up_blocks[i](x, skips[i], links[i]) if linked else self.up_blocks[i](x, skips[i])
            if linked and getattr(links[i], "planar_out", False):
                x = self.decode_blocks[i](x, planar=True)       # the concat as two planes (ops.SkipLink.planar_view)
            else:
                x = self.decode_blocks[i](x)
        ops._DROP_POOL.clear()
        return self._head(x)


# --------------------------------------------------------------------------- model zoo
def _stacks_by_level(level):
    return {'num_stacks': max(level, 1)}


class _UnetWrapper(nn.Module):
    def forward(self, x):
        return self.net(x)


class ResUnet3D(_UnetWrapper):
    """Residual 3D U-Net: ResBlockStack encoders (max(level,1) blocks), stride-2 ResBlock pooling,
    ConvTrans3D up-sampling + concat, ResBlock decoders.  This is the model every training script uses."""

    def __init__(self, num_pool=4, num_features=30, in_channels=1, out_channels=1):
        super().__init__()
        self.num_pool = num_pool
        self.num_features = num_features
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.net = Unet(in_channels=in_channels, out_channels=out_channels,
                        paired_features=generate_paired_features(num_pool, num_features),
                        pool_block=ResBlock, pool_kwargs={'stride': 2},
                        encode_block=ResBlockStack, encode_kwargs_fn=_stacks_by_level,
                        decode_block=ResBlock)


class ResAttrUnet3D(_UnetWrapper):
    """ResUnet3D with attention-gated skips (the gate runs on the conv kernels + ru3d_pointwise: ops.AttGateFn)."""

    def __init__(self, num_pool=4, num_features=30, in_channels=1, out_channels=1):
        super().__init__()
        self.num_pool = num_pool
        self.num_features = num_features
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.net = Unet(in_channels=in_channels, out_channels=out_channels,
                        paired_features=generate_paired_features(num_pool, num_features),
                        pool_block=ResBlock, pool_kwargs={'stride': 2}, up_kwargs={'attention': True},
                        encode_block=ResBlockStack, encode_kwargs_fn=_stacks_by_level,
                        decode_block=ResBlock)


class ResAttrUnet3D2(_UnetWrapper):
    """Fixed 30/60/120/240/320 channel plan with attention-gated skips."""

    def __init__(self, in_channels=1, out_channels=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        widths = [30, 60, 120, 240, 320]
        plan = [[c, c] for c in widths] + [[320, 320]] + [[c, c] for c in reversed(widths)]
        self.net = Unet(in_channels=in_channels, out_channels=out_channels, paired_features=plan,
                        pool_block=ResBlock, pool_kwargs={'stride': 2}, up_kwargs={'attention': True},
                        encode_block=ResBlockStack, encode_kwargs_fn=_stacks_by_level,
                        decode_block=ResBlock)


class ResAttrBNUnet3D(_UnetWrapper):
    """BatchNorm + attention variant, on the native kernels in both modes.  Training: batch-pooled statistics with the
    Dropout3d factors folded in, running averages, gamma / beta gradients (ops.ResBlockBNFn / UpBNFn; RU3D_BN_TRAIN=torch
    keeps the blocks torch modules as a cross-check).  Inference (eval + no_grad - how the reference's scripts use this
    net, as the coarse model of nb_post_iia.py:20): BatchNorm with running statistics is a per-channel affine map."""

    def __init__(self, num_pool=4, num_features=30, in_channels=1, out_channels=1):
        super().__init__()
        self.num_pool = num_pool
        self.num_features = num_features
        self.in_channels = in_channels
        self.out_channels = out_channels
        bn = {'norm_op': nn.BatchNorm3d}
        self.net = Unet(in_channels=in_channels, out_channels=out_channels,
                        paired_features=generate_paired_features(num_pool, num_features),
                        pool_block=ResBlock, pool_kwargs={'stride': 2, **bn},
                        up_kwargs={'attention': True, **bn},
                        encode_block=ResBlockStack, encode_kwargs=bn, encode_kwargs_fn=_stacks_by_level,
                        decode_block=ResBlock, decode_kwargs=bn)
