"""Drop-in `transform` module: the reference's case transforms (reference transform.py) for host-side DataLoader
workers, plus the on-device pipeline (`DeviceAugment`, augment.py) that replaces them when the cases live in HBM.

Every class / function of the reference module that the training and inference scripts import is here under its
name and with its arguments (nb_train_iia.py:8-10, data.py:1-2, trainer.py:8): rescale, resize, crop_pad_to_bbox,
pad, crop_pad, to_tensor / to_numpy, to_one_hot, combination_labels, remove_small_region and the Random* / Crop* /
To* classes; `Compose` stands in for torchvision's (absent here).  Cases are dicts with a channels-last float32
'image' [d1, d2, d3, C] and an integer 'label' [d1, d2, d3]; the interpolation is scipy.ndimage.zoom exactly as in the
reference (third-party there too).  The random draws come from numpy's global generator in the reference's order, so a
seed reproduces the reference's patches (checked against tests/golden/g7_augment.npz).
"""
import numpy as np
import scipy.ndimage as ndi

from augment import DeviceAugment, DeviceCase  # noqa: F401  (the on-device replacement of the Random* chain)


class Compose(object):
    """torchvision.transforms.Compose as the scripts use it (nb_train_iia.py:30)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, case):
        for t in self.transforms:
            case = t(case)
        return case


def _as_range(v):
    if isinstance(v, float):
        assert 0 <= v <= 1, "If range is a single number, it must be non negative"
        return [1 - v, 1 + v]
    return v


def _per_axis(v, dim):
    return list(v) if isinstance(v, (np.ndarray, tuple, list)) else [v] * dim


# ------------------------------------------------------------------ layout helpers (transform.py:23-30, 144-173)
def split_dim(input, axis=-1):
    return [np.squeeze(a, axis=axis) for a in np.split(input, input.shape[axis], axis=axis)]


def slice_dim(input, slice, axis=-1):
    return split_dim(input, axis=axis)[slice]


def to_tensor(input):
    """(d1, ..., dn, C) -> (C, d1, ..., dn)"""
    return np.moveaxis(input, -1, 0)


def to_numpy(input):
    """(C, d1, ..., dn) -> (d1, ..., dn, C)"""
    return np.moveaxis(input, 0, -1)


def to_one_hot(input, num_classes, to_tensor=False):
    """transform.py:262-276: labels -> one-hot in the label's dtype, class axis last (or first)."""
    onehot = np.eye(num_classes)[input]
    if to_tensor:
        onehot = np.moveaxis(onehot, -1, 0)
    return onehot.astype(input.dtype)


# ------------------------------------------------------------------ resampling (transform.py:32-101)
def _zoom(a, scale, order, mode, cval):
    return ndi.zoom(a.astype(np.float32), scale, order=order, mode=mode, cval=cval)


def rescale(input, scale, order=1, mode='reflect', cval=0, is_label=False, multi_class=False):
    """scipy.ndimage.zoom with label support: labels with three or more classes present are interpolated per class
    (one-hot) and arg-maxed; fewer classes (or order 0) go through zoom directly and are cast back."""
    dtype = input.dtype
    if is_label:
        num_classes = np.unique(input).max() + 1
    if order == 0 or not is_label or num_classes < 3:
        if multi_class:
            planes = np.array([_zoom(c, scale, order, mode, cval) for c in to_tensor(input)])
            return to_numpy(planes).astype(dtype)
        return _zoom(input, scale, order, mode, cval).astype(dtype)
    planes = np.array([_zoom(c, scale, order, mode, cval) for c in to_one_hot(input, num_classes, to_tensor=True)])
    return np.argmax(planes, axis=0).astype(dtype)


def resize(input, shape, order=1, mode='reflect', cval=0, is_label=False):
    orig = input.shape
    multi_class = len(shape) == len(orig) - 1
    scale = np.array(shape) / np.array(orig[:len(shape)])
    return rescale(input, scale, order=order, mode=mode, cval=cval, is_label=is_label, multi_class=multi_class)


# ------------------------------------------------------------------ cropping / padding (transform.py:387-437)
def gen_bbox_for_crop(crop_size, orig_shape, crop_margin, crop_mode):
    assert crop_mode == "center" or crop_mode == "random", "crop mode must be either center or random"
    bbox = []
    for i in range(len(orig_shape)):
        if i >= len(crop_size):
            bbox.append([0, orig_shape[i]])
            continue
        room = orig_shape[i] - crop_size[i] - crop_margin[i]
        if crop_mode == 'random' and room > crop_margin[i]:
            lo = np.random.randint(crop_margin[i], room)
        else:
            lo = (orig_shape[i] - crop_size[i]) // 2
        bbox.append([lo, lo + crop_size[i]])
    return bbox


def crop_pad_to_bbox(input, bbox, pad_mode='constant', pad_cval=0):
    shape = input.shape
    inside = tuple(slice(max(0, bbox[d][0]), min(bbox[d][1], shape[d])) for d in range(len(shape)))
    out = input[inside]
    widths = [[abs(min(0, bbox[d][0])), abs(min(0, shape[d] - bbox[d][1]))] for d in range(len(shape))]
    if any(w > 0 for pair in widths for w in pair):
        out = np.pad(out, widths, pad_mode, constant_values=pad_cval)
    return out.astype(input.dtype)


def crop_pad(input, crop_size, crop_mode='center', crop_margin=0, pad_mode='constant', pad_cval=0):
    margin = _per_axis(crop_margin, len(crop_size))
    return crop_pad_to_bbox(input, gen_bbox_for_crop(crop_size, input.shape, margin, crop_mode), pad_mode, pad_cval)


def pad(input, pad_size, pad_mode='constant', pad_cval=0):
    size = [max(input.shape[d], pad_size[d]) for d in range(len(pad_size))]
    return crop_pad(input, size, pad_mode=pad_mode, pad_cval=pad_cval)


# ------------------------------------------------------------------ labels (transform.py:5-20, 323-384)
def remove_small_region(input, threshold):
    labels, _ = ndi.label(input)
    areas = np.bincount(labels.ravel())
    input[(areas < threshold)[labels]] = 0
    return input


def combination_labels(input, combinations, num_classes):
    """transform.py:323-363: merge label classes.  `combinations` is one group or a list of groups of class indices.
    The new class order follows the old classes 0, 1, ...: a class that belongs to a group puts that whole group at
    its place (once), a class in no group stays a class of its own; voxels get the index of their (merged) class."""
    groups = [list(combinations)] if np.ndim(combinations[0]) == 0 else [list(g) for g in combinations]
    order, placed = [], set()
    for c in range(num_classes):
        owners = [i for i, g in enumerate(groups) if c in g]
        if not owners:
            order.append([c])
        for i in owners:
            if i not in placed:
                order.append(groups[i])
                placed.add(i)
    onehot = to_one_hot(input, num_classes, to_tensor=True)
    planes = np.array([np.any([onehot[c].astype(bool) for c in group], axis=0) for group in order])
    return np.argmax(planes, axis=0).astype(input.dtype)


# ------------------------------------------------------------------ intensity (transform.py:176-193)
def adjust_contrast(input, factor):
    mean = input.mean()
    return ((input - mean) * factor + mean).astype(input.dtype)


def adjust_brightness(input, factor):
    low = input.min()
    return ((input - low) * factor + low).astype(input.dtype)


def adjust_gamma(input, gamma, epsilon=1e-7):
    low, high = input.min(), input.max()
    span = high - low + epsilon
    return (np.power((input - low) / span, gamma) * span + low).astype(input.dtype)


# ------------------------------------------------------------------ transform classes
class _ImageFactor(object):
    """One uniform draw from a range, applied to case['image'] (RandomContrast / Brightness / Gamma: :196-259)."""
    fn = None

    def __init__(self, factor_range):
        self.factor_range = _as_range(factor_range)

    def __call__(self, case):
        factor = np.random.uniform(self.factor_range[0], self.factor_range[1])
        case['image'] = type(self).fn(case['image'], factor)
        return case


class RandomContrast(_ImageFactor):
    fn = staticmethod(adjust_contrast)


class RandomBrightness(_ImageFactor):
    fn = staticmethod(adjust_brightness)


class RandomGamma(_ImageFactor):
    fn = staticmethod(adjust_gamma)

    def __init__(self, gamma_range):
        super().__init__(gamma_range)
        self.gamma_range = self.factor_range


class RandomMirror(object):
    def __init__(self, p_per_axis):
        self.p_per_axis = p_per_axis

    def __call__(self, case):
        self.p_per_axis = _per_axis(self.p_per_axis, len(case['image'].shape) - 1)
        for axis, p in enumerate(self.p_per_axis):
            if np.random.uniform() < p:
                case['image'] = np.flip(case['image'], axis).copy()
                case['label'] = np.flip(case['label'], axis).copy()
        return case


class ToTensor(object):
    def __call__(self, case):
        case['image'] = to_tensor(case['image'])
        return case


class ToNumpy(object):
    def __call__(self, case):
        case['image'] = to_numpy(case['image'])
        return case


class ToOnehot(object):
    def __init__(self, num_classes, to_tensor=False):
        self.num_classes, self.to_tensor = num_classes, to_tensor

    def __call__(self, case):
        case['label'] = to_one_hot(case['label'], self.num_classes, self.to_tensor)
        return case


class CombineLabels(object):
    def __init__(self, combinations, num_classes):
        self.combinations, self.num_classes = combinations, num_classes

    def __call__(self, case):
        case['label'] = combination_labels(case['label'], self.combinations, self.num_classes)
        return case


class RemoveSmallRegion(object):
    def __init__(self, threshold):
        self.threshold = threshold

    def __call__(self, case):
        case['label'] = remove_small_region(case['label'], self.threshold)
        return case


class Resize(object):
    def __init__(self, shape):
        self.shape = shape

    def __call__(self, case):
        case['image'] = resize(case['image'], self.shape)
        case['label'] = resize(case['label'], self.shape, is_label=True)
        return case


class RandomRescale(object):
    def __init__(self, scale):
        self.scale = _as_range(scale)

    def __call__(self, case):
        s = np.random.uniform(self.scale[0], self.scale[1])
        case['image'] = rescale(case['image'], s)
        case['label'] = rescale(case['label'], s, is_label=True)
        return case


class Crop(object):
    """transform.py:440-511: crop image and label with one box; retry until every enforce_label_indices entry is in
    the cropped label."""

    def __init__(self, crop_size=128, crop_mode='center', crop_margin=0, enforce_label_indices=[],
                 image_pad_mode='constant', image_pad_cval=0, label_pad_mode='constant', label_pad_cval=0):
        self.crop_size, self.crop_mode, self.crop_margin = crop_size, crop_mode, crop_margin
        self.enforce_label_indices = ([enforce_label_indices] if isinstance(enforce_label_indices, int)
                                      else enforce_label_indices)
        self.image_pad_mode, self.image_pad_cval = image_pad_mode, image_pad_cval
        self.label_pad_mode, self.label_pad_cval = label_pad_mode, label_pad_cval

    def _box(self, image, label, size):
        while True:
            bbox = gen_bbox_for_crop(size, image.shape, self.crop_margin, self.crop_mode)
            cropped = crop_pad_to_bbox(label, bbox[:-1], self.label_pad_mode, self.label_pad_cval)
            present = np.unique(cropped)
            if all(i in present for i in self.enforce_label_indices):
                return bbox, cropped

    def __call__(self, case):
        dim = len(case['image'].shape) - 1
        self.crop_size = _per_axis(self.crop_size, dim)
        self.crop_margin = _per_axis(self.crop_margin, dim)
        bbox, cropped_label = self._box(case['image'], case['label'], self.crop_size)
        case['image'] = crop_pad_to_bbox(case['image'], bbox, self.image_pad_mode, self.image_pad_cval)
        case['label'] = cropped_label
        return case


class RandomCrop(Crop):
    def __init__(self, crop_size=128, crop_margin=0, enforce_label_indices=[], image_pad_mode='constant',
                 image_pad_cval=0, label_pad_mode='constant', label_pad_cval=0):
        super().__init__(crop_size, crop_mode='random', crop_margin=crop_margin,
                         enforce_label_indices=enforce_label_indices, image_pad_mode=image_pad_mode,
                         image_pad_cval=image_pad_cval, label_pad_mode=label_pad_mode, label_pad_cval=label_pad_cval)


class CenterCrop(Crop):
    def __init__(self, crop_size=128, image_pad_mode='constant', image_pad_cval=0, label_pad_mode='constant',
                 label_pad_cval=0):
        super().__init__(crop_size, crop_mode='center', image_pad_mode=image_pad_mode, image_pad_cval=image_pad_cval,
                         label_pad_mode=label_pad_mode, label_pad_cval=label_pad_cval)


class RandomRescaleCrop(Crop):
    """transform.py:573-652: draw a scale, crop round(size / scale), resize the crop to `size`."""

    def __init__(self, scale, crop_size=128, crop_mode='center', crop_margin=0, enforce_label_indices=[],
                 image_pad_mode='constant', image_pad_cval=0, label_pad_mode='constant', label_pad_cval=0):
        super().__init__(crop_size, crop_mode=crop_mode, crop_margin=crop_margin,
                         enforce_label_indices=enforce_label_indices, image_pad_mode=image_pad_mode,
                         image_pad_cval=image_pad_cval, label_pad_mode=label_pad_mode, label_pad_cval=label_pad_cval)
        self.scale = _as_range(scale)

    def __call__(self, case):
        dim = len(case['image'].shape) - 1
        self.crop_size = _per_axis(self.crop_size, dim)
        self.crop_margin = _per_axis(self.crop_margin, dim)
        s = np.random.uniform(self.scale[0], self.scale[1])
        before = np.round(np.array(self.crop_size) / s).astype(int)       # the reference's np.int (gone in numpy 1.24)
        bbox, cropped_label = self._box(case['image'], case['label'], before)
        cropped_image = crop_pad_to_bbox(case['image'], bbox, self.image_pad_mode, self.image_pad_cval)
        case['image'] = resize(cropped_image, self.crop_size)
        case['label'] = resize(cropped_label, self.crop_size, is_label=True)
        return case
