"""CPU oracle for the 3D U-Net training hot path.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch CPU restatement (torch-CPU functional ops + numpy) of the
algorithm of the reference hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path (the HIP kernels
behind `include/ru3d.h`) never routes through it and fails loudly when the HIP
library is missing.

Parity pin: every function here is checked in `tests/test_oracle_golden.py` against
the committed fixtures `tests/golden/g1..g5`, which were produced by importing the
reference's own `network.py` / `loss.py` (see `tests/golden/make_golden.py`).

Where the arithmetic lives: the reference delegates conv/norm arithmetic to PyTorch
(ATen/oneDNN, unpinned third-party dependency); the restatement calls the same
primitive library for convolutions on CPU and additionally carries an independent
numpy direct-convolution (`conv3d_naive`, `conv_transpose3d_naive`) that pins the
primitive itself on small cases.

Reference behaviour followed (file:line in /root/reference):
  network.py:104-141  ResUnet3D / generate_paired_features  -> `paired_features`, `unet_forward`
  network.py:298-320  ConvTrans3D (convT k3 s2 p1 -> far-side zero pad -> IN -> LeakyReLU) -> `conv_trans`
  network.py:323-350  UpConcat (cat((up, skip), dim=1))     -> `up_concat`
  network.py:374-416  ResBlock                              -> `res_block`
  network.py:419-449  ResBlockStack                         -> `res_stack`
  network.py:470-565  Unet assembler / forward order        -> `unet_forward`
  loss.py:7-48        logits / flatten / dice               -> `class_sums`, `tversky`
  loss.py:51-82,169   focal_loss / FocalLoss                -> `focal_loss`
  loss.py:85-166      Dice / DiceLoss                       -> `dice_metric`, `dice_loss`
  loss.py:196-254     HybirdLoss                            -> `hybird_loss`
"""
import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.01   # nn.LeakyReLU default (network.py:386)
IN_EPS = 1e-5        # nn.InstanceNorm3d default (network.py:384)
DROP_P = 0.5         # nn.Dropout3d(p=0.5) (network.py:383)


# --------------------------------------------------------------------------- storage-precision model
class _Store(torch.autograd.Function):
    """Models a tensor that is STORED in a narrower dtype between kernels (bf16 mode of the HIP path):
    the forward value and the gradient flowing back through the same point are both rounded."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


_STORAGE = [None]   # None = exact (fp32/fp64) pipeline


def set_storage(dtype):
    """oracle.set_storage(torch.bfloat16) makes every inter-kernel tensor (and every weight as the kernels
    read it) round-trip through bf16, with fp32+ accumulation inside each op - a statistical model of the
    bf16 mode, used to bound its error; set_storage(None) restores the exact pipeline."""
    _STORAGE[0] = dtype


def _st(x):
    return x if _STORAGE[0] is None else _Store.apply(x, _STORAGE[0])


def _wq(w):
    return w if _STORAGE[0] is None else _Store.apply(w, _STORAGE[0])


# --------------------------------------------------------------------------- model
def paired_features(num_pool, num_features):
    """network.py:135-141."""
    down = [[num_features * 2 ** i] * 2 for i in range(num_pool)]
    bottom = [[num_features * 2 ** num_pool] * 2]
    up = [[num_features * 2 ** i] * 2 for i in range(num_pool - 1, -1, -1)]
    return down + bottom + up


def instance_norm(x, scale=None):
    """InstanceNorm3d(affine=False, track_running_stats=False): biased variance over D*H*W.

    `scale` is the optional per-(n,c) Dropout3d factor (0 or 1/(1-p)) applied BEFORE the norm
    (network.py:412-414: conv1 -> dropout -> norm).
    """
    if scale is not None:
        x = x * scale[:, :, None, None, None]
    mean = x.mean(dim=(2, 3, 4), keepdim=True)
    var = x.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + IN_EPS)


def lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def res_block(x, w, prefix, stride=1, keep=None):
    """network.py:405-416.  `keep`: optional [N, Cout] 0/1 keep mask for the Dropout3d site."""
    w1, b1 = w[prefix + "conv1.weight"], w[prefix + "conv1.bias"]
    w2, b2 = w[prefix + "conv2.weight"], w[prefix + "conv2.bias"]
    cin, cout = w1.shape[1], w1.shape[0]
    if cin != cout or stride != 1:
        skip = _st(F.conv3d(x, _wq(w[prefix + "skip_conv.weight"]), w[prefix + "skip_conv.bias"], stride=stride))
    else:
        skip = x
    y = _st(F.conv3d(x, _wq(w1), b1, stride=stride, padding=1))
    scale = None if keep is None else keep.to(y.dtype) / (1.0 - DROP_P)
    y = _st(lrelu(instance_norm(y, scale)))
    y = _st(F.conv3d(y, _wq(w2), b2, padding=1))
    return _st(lrelu(instance_norm(y) + skip))


def res_stack(x, w, prefix, num_stacks, stride=1, keeps=None):
    """network.py:446-449."""
    for j in range(num_stacks):
        k = None if keeps is None else keeps.get(prefix + "res_blocks.%d.dropout" % j)
        x = res_block(x, w, prefix + "res_blocks.%d." % j, stride if j == 0 else 1, k)
    return x


def conv_trans(x, w, prefix):
    """network.py:311-317: ConvTranspose3d(k3,s2,p1) -> ConstantPad3d((0,1,0,1,0,1)) -> IN -> LeakyReLU."""
    y = F.conv_transpose3d(x, _wq(w[prefix + "up.0.weight"]), w[prefix + "up.0.bias"], stride=2, padding=1)
    y = _st(F.pad(y, (0, 1, 0, 1, 0, 1), value=0.0))
    return _st(lrelu(instance_norm(y)))


def up_concat(x, skip, w, prefix):
    """network.py:346-350 (attention=False): up-sampled first, skip second."""
    return torch.cat((conv_trans(x, w, prefix + "conv_trans."), skip), dim=1)


def unet_forward(x, w, num_pool, keeps=None, prefix="net."):
    """network.py:549-565 with the ResUnet3D block choice (network.py:116-129).

    w: dict name -> tensor with the reference's state_dict keys.
    keeps: optional dict '<module path of the Dropout3d>' -> [N,C] keep mask; None = eval mode.
    """
    def k(name):
        return None if keeps is None else keeps.get(name)

    x = _st(F.conv3d(_st(x), _wq(w[prefix + "conv.weight"]), w[prefix + "conv.bias"], padding=1))
    skips = []
    for i in range(num_pool):
        x = res_stack(x, w, prefix + "encode_blocks.%d." % i, max(i, 1), 1, keeps)
        skips.append(x)
        x = res_block(x, w, prefix + "pool_blocks.%d." % i, 2, k(prefix + "pool_blocks.%d.dropout" % i))
    x = res_stack(x, w, prefix + "encode_blocks.%d." % num_pool, max(num_pool, 1), 1, keeps)
    for i in range(num_pool - 1, -1, -1):
        x = up_concat(x, skips[i], w, prefix + "up_blocks.%d." % i)
        x = res_block(x, w, prefix + "decode_blocks.%d." % i, 1, k(prefix + "decode_blocks.%d.dropout" % i))
    return F.conv3d(x, _wq(w[prefix + "fc.weight"]), w[prefix + "fc.bias"])   # logits stay fp32


def init_state_dict(num_pool, num_features, in_channels, out_channels, seed=0):
    """Deterministic state_dict with the reference's key names/shapes and PyTorch-default init
    distributions (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both
    weight and bias).  Used on the GPU box where the reference ctor is not available; the
    *values* differ from `torch.manual_seed(0); ResUnet3D(...)` (creation order), which only
    matters for fixtures - and those carry their weights explicitly."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def conv(name, cout, cin, k, transposed=False):
        shape = (cin, cout, k, k, k) if transposed else (cout, cin, k, k, k)
        fan_in = shape[1] * k ** 3
        bound = 1.0 / fan_in ** 0.5
        w[name + ".weight"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        w[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def res(prefix, cin, cout):
        conv(prefix + "conv1", cout, cin, 3)
        conv(prefix + "conv2", cout, cout, 3)
        conv(prefix + "skip_conv", cout, cin, 1)

    Fc = [num_features * 2 ** i for i in range(num_pool + 1)]
    for i in range(num_pool):
        res("net.pool_blocks.%d." % i, Fc[i], Fc[i + 1])
    for i in range(num_pool):
        conv("net.up_blocks.%d.conv_trans.up.0" % i, Fc[i], Fc[i + 1], 3, transposed=True)
    for i in range(num_pool + 1):
        for j in range(max(i, 1)):
            res("net.encode_blocks.%d.res_blocks.%d." % (i, j), Fc[i], Fc[i])
    for i in range(num_pool):
        res("net.decode_blocks.%d." % i, 2 * Fc[i], Fc[i])
    conv("net.conv", Fc[0], in_channels, 3)
    conv("net.fc", out_channels, Fc[0], 1)
    return w


def unused_param_keys(w, prefix="net."):
    """Parameters that never receive a gradient: skip_conv of ResBlocks with in==out, stride 1
    (network.py:403 constructs it always, :406-409 uses it conditionally)."""
    out = []
    for k in w:
        if k.endswith("skip_conv.weight"):
            base = k[: -len("skip_conv.weight")]
            w1 = w[base + "conv1.weight"]
            is_pool = ".pool_blocks." in k
            if w1.shape[0] == w1.shape[1] and not is_pool:
                out += [k, base + "skip_conv.bias"]
    return out


# --------------------------------------------------------------------------- loss
def _norm_weight(weight_v, C):
    wv = torch.ones(C) if weight_v is None else torch.tensor(weight_v, dtype=torch.float32)
    # loss.py:69/155/237: weight = normalize(weight_v, p=1); weight_c and the class-presence
    # mask are computed and then overwritten (bug-compatible).
    return F.normalize(wv.float(), p=1, dim=0)


def _probs(logits):
    """loss.py:7-11 / 223-228: softmax over C, sigmoid when C == 1.  Returns (p, logp) as [N*V, C]."""
    N, C = logits.shape[:2]
    z = logits.reshape(N, C, -1).transpose(1, 2).reshape(-1, C)
    if C > 1:
        logp = F.log_softmax(z, -1)
        return logp.exp(), logp
    p = torch.sigmoid(z)
    return p, torch.log(p)


def _one_hot(target, C):
    """loss.py:27: F.one_hot(target, C) (raises when a label >= C, e.g. C==1 with labels {0,1})."""
    return F.one_hot(target.reshape(-1).long(), num_classes=C)


def tversky(p, g, alpha=0.5, beta=0.5, smooth=1e-7):
    """loss.py:32-48 on flat vectors."""
    p = p.reshape(-1)
    g = g.reshape(-1)
    tp = (p * g).sum()
    fn = ((1 - p) * g).sum()
    fp = (p * (1 - g)).sum()
    return (tp + smooth) / (tp + alpha * fn + beta * fp + smooth)


def dice_metric(logits, target, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
    """loss.py:104-120."""
    C = logits.shape[1]
    w = _norm_weight(weight_v, C)
    p = torch.softmax(logits, 1) if C > 1 else torch.sigmoid(logits)
    p = p.reshape(p.shape[0], C, -1).transpose(1, 2).reshape(-1, C)
    g = _one_hot(target, C)
    d = torch.stack([tversky(p[:, i], g[:, i], alpha, beta, smooth) for i in range(C)])
    return (w * d).sum()


def dice_loss(logits, target, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
    """loss.py:143-166."""
    C = logits.shape[1]
    w = _norm_weight(weight_v, C)
    p = torch.softmax(logits, 1) if C > 1 else torch.sigmoid(logits)
    p = p.reshape(p.shape[0], C, -1).transpose(1, 2).reshape(-1, C)
    g = _one_hot(target, C)
    d = torch.stack([tversky(p[:, i], g[:, i], alpha, beta, smooth) for i in range(C)])
    return (w * (1 - d)).sum()


def focal_loss(logits, target, gamma=2, weight_v=None):
    """loss.py:51-82 via FocalLoss.forward (loss.py:187-193)."""
    C = logits.shape[1]
    w = _norm_weight(weight_v, C)
    p, logp = _probs(logits)
    g = _one_hot(target, C)
    focals = -(1 - p) ** gamma * g * logp
    return (w * (C * focals.mean(dim=0))).sum()


def hybird_loss(logits, target, gamma=2, weight_v=None, alpha=0.5, beta=0.5, smooth=1e-7):
    """loss.py:218-254."""
    C = logits.shape[1]
    w = _norm_weight(weight_v, C)
    p, logp = _probs(logits)
    g = _one_hot(target, C)
    focals = C * (-(1 - p) ** gamma * g * logp).mean(dim=0)
    d = torch.stack([tversky(p[:, i], g[:, i], alpha, beta, smooth) for i in range(C)])
    return (w * (1 - d + focals)).sum()


# --------------------------------------------------------------------------- train step
def train_step(w, x, y, num_pool, loss_kwargs=None, keeps=None):
    """One forward + HybirdLoss + backward (trainer.py:480-495).  Returns (loss, logits, grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    logits = unet_forward(x, leaves, num_pool, keeps)
    loss = hybird_loss(logits, y, **(loss_kwargs or {}))
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return loss.detach(), logits.detach(), grads


def adam_step(w, grads, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (nb_train_iia.py:18), restated; parameters without a gradient are skipped."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    b1, b2 = betas
    for k, g in grads.items():
        m = state.setdefault("m/" + k, torch.zeros_like(g))
        v = state.setdefault("v/" + k, torch.zeros_like(g))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
        w[k] = w[k] - (lr / (1 - b1 ** t)) * (m / denom)
    return w


# --------------------------------------------------------------------------- independent primitive pins
def conv3d_naive(x, w, b=None, stride=1, padding=0):
    """Direct 3-D cross-correlation in numpy float64 (pins F.conv3d on small cases).
    x [N,Cin,D,H,W], w [Cout,Cin,k,k,k]."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    N, Cin, D, H, W = x.shape
    Cout, _, k, _, _ = w.shape
    xp = np.pad(x, ((0, 0), (0, 0)) + ((padding, padding),) * 3)
    Do = (D + 2 * padding - k) // stride + 1
    Ho = (H + 2 * padding - k) // stride + 1
    Wo = (W + 2 * padding - k) // stride + 1
    out = np.zeros((N, Cout, Do, Ho, Wo))
    for kd in range(k):
        for kh in range(k):
            for kw in range(k):
                patch = xp[:, :, kd:kd + stride * Do:stride, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride]
                out += np.einsum("ncdhw,oc->nodhw", patch, w[:, :, kd, kh, kw])
    if b is not None:
        out += np.asarray(b, dtype=np.float64)[None, :, None, None, None]
    return out


def conv_transpose3d_naive(x, w, b=None, stride=2, padding=1):
    """Direct transposed conv in numpy float64.  x [N,Cin,D,H,W], w [Cin,Cout,k,k,k]."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    N, Cin, D, H, W = x.shape
    _, Cout, k, _, _ = w.shape
    full = np.zeros((N, Cout, (D - 1) * stride + k, (H - 1) * stride + k, (W - 1) * stride + k))
    for kd in range(k):
        for kh in range(k):
            for kw in range(k):
                contrib = np.einsum("ncdhw,co->nodhw", x, w[:, :, kd, kh, kw])
                full[:, :, kd:kd + stride * D:stride, kh:kh + stride * H:stride, kw:kw + stride * W:stride] += contrib
    p = padding
    out = full[:, :, p:full.shape[2] - p, p:full.shape[3] - p, p:full.shape[4] - p]
    if b is not None:
        out = out + np.asarray(b, dtype=np.float64)[None, :, None, None, None]
    return out


# --------------------------------------------------------------------------- synthetic data
def synth_image(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).clamp_(-2.34, 2.64)


def phantom_labels(n, dims, num_classes):
    d, h, w = dims
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    lab = np.zeros((n, d, h, w), dtype=np.int64)
    for i in range(n):
        cz, cy, cx = d * (0.5 + 0.05 * i), h * 0.5, w * (0.45 + 0.05 * i)
        e = ((zz - cz) / (0.30 * d)) ** 2 + ((yy - cy) / (0.25 * h)) ** 2 + ((xx - cx) / (0.22 * w)) ** 2
        lab[i][e <= 1.0] = 1
        if num_classes > 2:
            s = (zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2
            lab[i][s <= (0.09 * min(dims)) ** 2] = 2
        if num_classes > 3:
            s = (zz - cz - 0.15 * d) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2
            lab[i][s <= (0.06 * min(dims)) ** 2] = 3
    return torch.from_numpy(lab)


# --------------------------------------------------------------------------- sliding-window inference
def center_crop_pad(a, size, cval=0):
    """transform.py:392-432 (crop_pad, crop_mode 'center', constant padding) on the leading len(size) axes."""
    a = np.asarray(a)
    box = []
    for i in range(a.ndim):
        if i < len(size):
            lo = (a.shape[i] - size[i]) // 2
            box.append((lo, lo + size[i]))
        else:
            box.append((0, a.shape[i]))
    cut = a[tuple(slice(max(0, lo), min(hi, a.shape[i])) for i, (lo, hi) in enumerate(box))]
    width = [(abs(min(0, lo)), abs(min(0, a.shape[i] - hi))) for i, (lo, hi) in enumerate(box)]
    if any(v > 0 for pair in width for v in pair):
        cut = np.pad(cut, width, "constant", constant_values=cval)
    return cut.astype(a.dtype)


def predict_per_patch(image, w, num_pool, num_classes=3, patch_size=(96, 96, 96), step_per_patch=4, one_hot=False):
    """trainer.py:17-98 with the model call replaced by `unet_forward(., w, num_pool)`.

    image: numpy [X, Y, Z, C].  `np.int` (removed in numpy 1.24) was the builtin int, so the centres are
    np.arange(..., dtype=int) exactly as the reference computes them (trainer.py:38-40)."""
    orig = image.shape[:3]
    image = center_crop_pad(image, [max(image.shape[d], patch_size[d]) for d in range(3)])     # transform.pad
    full = image.shape[:3]
    start = np.array([p // 2 for p in patch_size])
    end = np.array([full[i] - patch_size[i] // 2 for i in range(3)])
    num_steps = np.ceil([(end[i] - start[i]) / (patch_size[i] / step_per_patch) for i in range(3)])
    step = np.array([(end[i] - start[i]) / (num_steps[i] + 1e-8) for i in range(3)])
    step[step == 0] = 9999999
    axes = [np.arange(start[i], end[i] + 1e-8, step[i], dtype=int) for i in range(3)]
    result = torch.zeros([num_classes] + list(full))
    result_n = torch.zeros_like(result)
    x = torch.from_numpy(np.moveaxis(image, -1, 0)[None].astype(np.float32))                    # to_tensor
    with torch.no_grad():
        for cx in axes[0]:
            for cy in axes[1]:
                for cz in axes[2]:
                    win = (slice(cx - patch_size[0] // 2, cx + patch_size[0] // 2),
                           slice(cy - patch_size[1] // 2, cy + patch_size[1] // 2),
                           slice(cz - patch_size[2] // 2, cz + patch_size[2] // 2))
                    out = unet_forward(x[(slice(None), slice(None)) + win], w, num_pool)
                    out = torch.sigmoid(out) if num_classes == 1 else torch.softmax(out, dim=1)
                    result[(slice(None),) + win] += out[0]
                    result_n[(slice(None),) + win] += 1
    result = result / result_n
    if one_hot:
        result = np.moveaxis(result.numpy(), 0, -1).astype(np.float32)                          # to_numpy
    else:
        if num_classes == 1:
            result = torch.squeeze(result, dim=0)
        else:
            result = torch.argmax(torch.softmax(result, dim=0), dim=0)
        with np.errstate(invalid="ignore"):
            result = np.round(result.numpy()).astype(np.uint8)
    return center_crop_pad(result, orig), [a.tolist() for a in axes]
