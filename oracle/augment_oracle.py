"""CPU oracle of the patch sampling / augmentation path.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing in the product path).  A numpy restatement, without scipy, of what the reference's training
transform pipeline does (nb_train_iia.py:30-39); each function cites the reference lines it follows.  Pinned by
tests/golden/g7_augment.npz, which tests/golden/make_golden_augment.py produced by running the reference's own
transform.py.

The third-party piece is scipy.ndimage.zoom(order=1, mode='reflect') (no version pin in the reference; scipy 1.15
here): with the default grid_mode=False output sample o of an axis reads input coordinate o * (in - 1) / (out - 1)
(corner-aligned), interpolates linearly in float64 and returns the input's dtype; order 1 needs no spline prefilter.
"""
import numpy as np


def zoom_linear(a, out_shape):
    """scipy.ndimage.zoom(a.astype(float32), out_shape / a.shape, order=1, mode='reflect') restated: separable linear
    interpolation at corner-aligned coordinates, float64 arithmetic, float32 result.  (reference transform.py:61-65)"""
    v = np.asarray(a, dtype=np.float64)
    for ax, n_out in enumerate(out_shape):
        n_in = v.shape[ax]
        if n_out > 1:
            c = np.arange(n_out, dtype=np.float64) * (float(n_in - 1) / float(n_out - 1))
        else:
            c = np.zeros(1, dtype=np.float64)
        i0 = np.floor(c).astype(np.int64)
        i0 = np.clip(i0, 0, n_in - 1)
        w = c - i0
        i1 = np.clip(i0 + 1, 0, n_in - 1)
        lo = np.take(v, i0, axis=ax)
        hi = np.take(v, i1, axis=ax)
        shp = [1] * v.ndim
        shp[ax] = n_out
        w = w.reshape(shp)
        v = lo * (1.0 - w) + hi * w
    return v.astype(np.float32)


def resize_image(img, shape):
    """transform.py:77-101 `resize` with a channel axis (multi_class=True): every channel zoomed on its own
    (transform.py:49-58), result cast back to the image dtype."""
    return np.stack([zoom_linear(img[..., c], shape) for c in range(img.shape[-1])], axis=-1).astype(img.dtype)


def resize_label(lab, shape):
    """transform.py:77-101 with is_label=True -> `rescale` :32-74: fewer than three classes present (max + 1 < 3):
    the label itself is interpolated and TRUNCATED back to its integer dtype (:60-65); otherwise one-hot per class,
    each class interpolated, argmax (first maximum) (:66-74)."""
    num_classes = int(np.unique(lab).max()) + 1
    if num_classes < 3:
        return zoom_linear(lab.astype(np.float32), shape).astype(lab.dtype)
    planes = np.stack([zoom_linear((lab == c).astype(np.float32), shape) for c in range(num_classes)], axis=0)
    return np.argmax(planes, axis=0).astype(lab.dtype)


def gen_bbox(crop_size, orig_shape, crop_margin, crop_mode, rng=np.random):
    """transform.py:403-419: one randint per cropped axis when there is room, else (and in 'center' mode) centred;
    axes beyond len(crop_size) (the channel axis) are taken whole."""
    bbox = []
    for i in range(len(orig_shape)):
        if i < len(crop_size):
            if crop_mode == "random" and orig_shape[i] - crop_size[i] - crop_margin[i] > crop_margin[i]:
                lo = int(rng.randint(crop_margin[i], orig_shape[i] - crop_size[i] - crop_margin[i]))
            else:
                lo = int((orig_shape[i] - crop_size[i]) // 2)
            bbox.append([lo, lo + int(crop_size[i])])
        else:
            bbox.append([0, orig_shape[i]])
    return bbox


def crop_pad_to_bbox(a, bbox, cval=0):
    """transform.py:422-437: crop to the part of the box inside the array, then constant-pad to the box."""
    shape = a.shape
    sl = tuple(slice(max(0, bbox[d][0]), min(bbox[d][1], shape[d])) for d in range(len(shape)))
    out = a[sl]
    pw = [[abs(min(0, bbox[d][0])), abs(min(0, shape[d] - bbox[d][1]))] for d in range(len(shape))]
    if any(p > 0 for pp in pw for p in pp):
        out = np.pad(out, pw, "constant", constant_values=cval)
    return out.astype(a.dtype)


def adjust_contrast(x, f):
    """transform.py:176-179 (float32 arithmetic: python-float factors are weak scalars)."""
    mean = x.mean()
    return ((x - mean) * np.float32(f) + mean).astype(x.dtype)


def adjust_brightness(x, f):
    """transform.py:182-185"""
    mn = x.min()
    return ((x - mn) * np.float32(f) + mn).astype(x.dtype)


def adjust_gamma(x, g, epsilon=1e-7):
    """transform.py:188-193"""
    mn, mx = x.min(), x.max()
    rng = mx - mn + np.float32(epsilon)
    return (np.power((x - mn) / rng, np.float32(g)) * rng + mn).astype(x.dtype)


def pipeline(image, label, patch, scale=0.1, crop_mode="random", crop_margin=0, enforce_label_indices=(),
             mirror_p=(0.5, 0.5, 0.5), contrast=0.1, brightness=0.1, gamma=0.1, rng=np.random):
    """nb_train_iia.py:30-39 in one function; draws from `rng` in the reference's order:
    RandomRescaleCrop (transform.py:606-652): scale, then per attempt one randint per axis with room;
    RandomMirror (:290-301): one uniform per axis; RandomContrast / Brightness / Gamma (:212-259): one uniform each;
    ToTensor (:156-163).  image [x,y,z,C] float32, label [x,y,z] integer -> ([C,x,y,z] float32, [x,y,z])."""
    dim = image.ndim - 1
    patch = list(patch)
    margin = list(crop_margin) if isinstance(crop_margin, (list, tuple, np.ndarray)) else [crop_margin] * dim
    rngs = [1 - scale, 1 + scale] if isinstance(scale, float) else list(scale)
    s = rng.uniform(rngs[0], rngs[1])
    before = np.round(np.array(patch) / s).astype(int)
    while True:
        bbox = gen_bbox(before, image.shape, margin, crop_mode, rng)
        cl = crop_pad_to_bbox(label, bbox[:-1])
        present = np.unique(cl)
        if all(i in present for i in enforce_label_indices):
            break
    ci = crop_pad_to_bbox(image, bbox)
    img = resize_image(ci, patch)
    lab = resize_label(cl, patch)
    after_crop = img.copy()
    for ax, p in enumerate(mirror_p):
        if rng.uniform() < p:
            img = np.flip(img, ax).copy()
            lab = np.flip(lab, ax).copy()
    after_mirror = img.copy()

    def factor(r):
        lo, hi = (1 - r, 1 + r) if isinstance(r, float) else r
        return rng.uniform(lo, hi)
    img = adjust_contrast(img, factor(contrast))
    img = adjust_brightness(img, factor(brightness))
    img = adjust_gamma(img, factor(gamma))
    return np.ascontiguousarray(np.moveaxis(img, -1, 0)), lab, after_crop, after_mirror
