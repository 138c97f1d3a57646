"""On-device patch sampling + augmentation: the reference's training transform chain as HIP kernels.

`DeviceAugment` replaces, for volumes that live in HBM,

    Compose([RandomRescaleCrop(scale, patch, crop_mode=..., crop_margin=..., enforce_label_indices=...),
             RandomMirror(p_per_axis), RandomContrast(c), RandomBrightness(b), RandomGamma(g), ToTensor()])

of the training scripts (reference nb_train_iia.py:30-39; classes transform.py:573-652, 279-301, 196-259, 156-163).
At ~10^8 voxels/s per GPU the reference's two DataLoader workers running scipy.ndimage.zoom cannot feed one device; here
a patch costs three small kernels (csrc/augment.hip) and never leaves the GPU.

The random numbers are drawn on the HOST from numpy's global generator in exactly the reference's order (one uniform for
the scale, one randint per axis with room for the crop corner - again for every retry of enforce_label_indices -, one
uniform per mirror axis, one uniform per intensity op), so `np.random.seed(s)` reproduces the CPU pipeline's patch.
"""
import ctypes

import numpy as np
import torch

import _native as N
from _native import check, ptr, stream


def _range(v):
    if isinstance(v, float):
        assert 0 <= v <= 1, "If range is a single number, it must be non negative"
        return [1 - v, 1 + v]
    return list(v)


class DeviceCase:
    """A case resident in HBM: image fp32 [X, Y, Z, C] (channels last, the reference's case layout, data.py) and label
    [X, Y, Z] uint8 / int64.  Built once per case; patches are cut from it on the device."""

    def __init__(self, image, label=None, device="cuda"):
        img = torch.as_tensor(np.ascontiguousarray(image) if isinstance(image, np.ndarray) else image)
        if img.dim() == 3:
            img = img[..., None]
        self.image = img.to(device=device, dtype=torch.float32).contiguous()
        self.label = None
        if label is not None:
            lab = torch.as_tensor(np.ascontiguousarray(label) if isinstance(label, np.ndarray) else label)
            if lab.dtype not in (torch.uint8, torch.int64):
                lab = lab.to(torch.int64)
            self.label = lab.to(device).contiguous()
            if tuple(self.label.shape) != tuple(self.image.shape[:3]):
                raise ValueError("label %s does not match image %s" % (tuple(self.label.shape), tuple(self.image.shape)))

    @property
    def shape(self):
        return tuple(self.image.shape)


class DeviceAugment:
    def __init__(self, scale=0.1, crop_size=128, crop_mode="random", crop_margin=0, enforce_label_indices=(),
                 image_pad_cval=0, label_pad_cval=0, mirror_p=(0.5, 0.5, 0.5), contrast=0.1, brightness=0.1,
                 gamma=0.1, rng=None):
        """Arguments as the reference classes take them; contrast / brightness / gamma = None switches the op off,
        mirror_p = None the mirror.  rng: an object with numpy's `uniform` / `randint` (default: numpy's global one)."""
        assert crop_mode in ("center", "random"), "crop mode must be either center or random"
        self.scale = _range(scale)
        self.crop_size = crop_size
        self.crop_mode = crop_mode
        self.crop_margin = crop_margin
        self.enforce = [enforce_label_indices] if isinstance(enforce_label_indices, int) else list(enforce_label_indices)
        self.image_pad_cval, self.label_pad_cval = float(image_pad_cval), int(label_pad_cval)
        self.mirror_p = mirror_p
        self.contrast = None if contrast is None else _range(contrast)
        self.brightness = None if brightness is None else _range(brightness)
        self.gamma = None if gamma is None else _range(gamma)
        self.rng = rng if rng is not None else np.random

    # ------------------------------------------------------------------ host side: the draws
    def _bbox(self, before, shape, margin):
        """transform.py:403-419 for the three spatial axes."""
        lo = []
        for i in range(3):
            if self.crop_mode == "random" and shape[i] - before[i] - margin[i] > margin[i]:
                lo.append(int(self.rng.randint(margin[i], shape[i] - before[i] - margin[i])))
            else:
                lo.append(int((shape[i] - before[i]) // 2))
        return lo

    def _presence(self, case, lo, before, mask):
        lo_a = (ctypes.c_int32 * 3)(*lo)
        be_a = (ctypes.c_int32 * 3)(*before)
        lab = case.label
        code = N.LABEL_U8 if lab.dtype == torch.uint8 else N.LABEL_I64
        x, y, z = lab.shape
        N.note_device(lab.device)
        check(N.lib.ru3d_augment_label_presence(ptr(lab), code, x, y, z, lo_a, be_a, self.label_pad_cval, ptr(mask),
                                                stream()), "augment_label_presence")

    def sample(self, case, out_image=None, out_label=None):
        """One augmented patch of `case` (a DeviceCase): (image fp32 [C, px, py, pz], label int64 [px, py, pz] or None),
        written into out_image / out_label when given (slices of a batch buffer)."""
        shape = case.shape
        patch = [self.crop_size] * 3 if not isinstance(self.crop_size, (list, tuple, np.ndarray)) \
            else [int(v) for v in self.crop_size]
        margin = [self.crop_margin] * 3 if not isinstance(self.crop_margin, (list, tuple, np.ndarray)) \
            else [int(v) for v in self.crop_margin]
        dev = case.image.device
        s = self.rng.uniform(self.scale[0], self.scale[1])
        before = [int(v) for v in np.round(np.array(patch) / s).astype(int)]
        mask = torch.empty(1, dtype=torch.int32, device=dev)
        while True:
            lo = self._bbox(before, shape, margin)
            if case.label is None:
                break
            self._presence(case, lo, before, mask)
            if not self.enforce:
                break
            bits = int(mask.item()) & 0xffffffff          # only the enforce_label_indices loop reads back
            if all((bits >> min(int(i), 31)) & 1 for i in self.enforce):
                break
        flips = [0, 0, 0]
        if self.mirror_p is not None:
            ps = self.mirror_p if isinstance(self.mirror_p, (list, tuple, np.ndarray)) else [self.mirror_p] * 3
            for i, p in enumerate(ps):
                if self.rng.uniform() < p:
                    flips[i] = 1
        pr = N.PatchParams()
        for i in range(3):
            pr.lo[i], pr.before[i], pr.patch[i], pr.flip[i] = lo[i], before[i], patch[i], flips[i]
        pr.image_cval, pr.label_cval = self.image_pad_cval, self.label_pad_cval
        pr.gamma_eps = 1e-7
        for name, rng_ in (("contrast", self.contrast), ("brightness", self.brightness), ("gamma", self.gamma)):
            if rng_ is not None:
                setattr(pr, "do_" + name, 1)
                setattr(pr, name, float(self.rng.uniform(rng_[0], rng_[1])))
        c = shape[3]
        if out_image is None:
            out_image = torch.empty((c,) + tuple(patch), dtype=torch.float32, device=dev)
        if out_label is None and case.label is not None:
            out_label = torch.empty(tuple(patch), dtype=torch.int64, device=dev)
        if not out_image.is_contiguous() or (out_label is not None and not out_label.is_contiguous()):
            raise N.Ru3dError("DeviceAugment: output slices must be contiguous")
        ws = N.workspace(N.lib.ru3d_augment_workspace_bytes(*patch), dev)
        lab = case.label
        code = N.LABEL_I64 if (lab is None or lab.dtype == torch.int64) else N.LABEL_U8
        N.note_device(dev)
        check(N.lib.ru3d_augment_patch(ptr(case.image), ptr(lab), code, shape[0], shape[1], shape[2], c,
                                       ctypes.byref(pr), ptr(mask) if lab is not None else None, ptr(out_image),
                                       ptr(out_label), ptr(ws), ws.numel(), stream()), "augment_patch")
        return out_image, out_label

    def __call__(self, case):
        """Transform interface: case = {'image': DeviceCase | array [X,Y,Z,C], 'label': ...} -> the reference's
        post-ToTensor dict with device tensors (use with num_workers=0: DataLoader workers cannot touch the GPU)."""
        dc = case["image"] if isinstance(case["image"], DeviceCase) else DeviceCase(case["image"], case.get("label"))
        img, lab = self.sample(dc)
        out = dict(case)
        out["image"], out["label"] = img, lab
        return out

    def batch(self, cases, batch_size):
        """`batch_size` patches drawn from `cases` (list of DeviceCase) with replacement, like the reference's
        RandomSampler(replacement=True) (trainer.py:549-551): {'image': [N, C, ...] fp32, 'label': [N, ...] int64}."""
        patch = [self.crop_size] * 3 if not isinstance(self.crop_size, (list, tuple, np.ndarray)) \
            else [int(v) for v in self.crop_size]
        dev = cases[0].image.device
        c = cases[0].shape[3]
        x = torch.empty((batch_size, c) + tuple(patch), dtype=torch.float32, device=dev)
        y = torch.empty((batch_size,) + tuple(patch), dtype=torch.int64, device=dev)
        for b in range(batch_size):
            case = cases[int(self.rng.randint(0, len(cases)))]
            self.sample(case, x[b], y[b])
        return {"image": x, "label": y}


# --------------------------------------------------------------------------- plain resampling (inference: predict_case)
def _resample(image, label, out_shape):
    """scipy.ndimage.zoom(order=1) of a whole device volume to `out_shape` with the augmentation kernel (crop box = the
    volume, no mirror, no intensity ops): image fp32 [X,Y,Z,C] -> fp32 [C,x,y,z]; label [X,Y,Z] -> int64 [x,y,z] by the
    reference's label rule (transform.py:47-74)."""
    ref = image if image is not None else label
    dev = ref.device
    x, y, z = (int(v) for v in ref.shape[:3])
    pr = N.PatchParams()
    for i, (b, p) in enumerate(zip((x, y, z), out_shape)):
        pr.lo[i], pr.before[i], pr.patch[i], pr.flip[i] = 0, b, int(p), 0
    pr.gamma_eps = 1e-7
    c = int(image.shape[3]) if image is not None else 1
    out_image = torch.empty((c,) + tuple(int(p) for p in out_shape), dtype=torch.float32, device=dev) \
        if image is not None else None
    out_label = mask = None
    code = N.LABEL_I64
    if label is not None:
        if label.dtype not in (torch.uint8, torch.int64):
            label = label.to(torch.int64)
        label = label.contiguous()
        code = N.LABEL_U8 if label.dtype == torch.uint8 else N.LABEL_I64
        out_label = torch.empty(tuple(int(p) for p in out_shape), dtype=torch.int64, device=dev)
        mask = torch.empty(1, dtype=torch.int32, device=dev)
        lo_a = (ctypes.c_int32 * 3)(0, 0, 0)
        be_a = (ctypes.c_int32 * 3)(x, y, z)
        N.note_device(dev)
        check(N.lib.ru3d_augment_label_presence(ptr(label), code, x, y, z, lo_a, be_a, 0, ptr(mask), stream()),
              "augment_label_presence")
    ws = N.workspace(N.lib.ru3d_augment_workspace_bytes(*[int(p) for p in out_shape]), dev)
    N.note_device(dev)
    check(N.lib.ru3d_augment_patch(ptr(image), ptr(label), code, x, y, z, c, ctypes.byref(pr), ptr(mask),
                                   ptr(out_image), ptr(out_label), ptr(ws), ws.numel(), stream()), "augment_patch")
    return out_image, out_label


def resample_image(image, out_shape):
    """fp32 device volume [X,Y,Z,C] -> [x,y,z,C] (channels last again), every channel zoomed with order 1."""
    out, _ = _resample(image.to(torch.float32).contiguous(), None, out_shape)
    return out.permute(1, 2, 3, 0).contiguous()


def resample_label(label, out_shape):
    """integer device volume [X,Y,Z] -> int64 [x,y,z] (one-hot / argmax rule for three or more classes)."""
    _, out = _resample(None, label, out_shape)
    return out
