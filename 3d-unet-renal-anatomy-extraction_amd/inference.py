"""Sliding-window inference on the native path (reference trainer.py:17-98, `predict_per_patch`).

Same call, same window placement and the same merge arithmetic as the reference, including what its
code does rather than what it intends:
  * the window centres come from `np.arange(start, end + 1e-8, step, dtype=np.int)` with a fractional
    step (trainer.py:38-40).  For an integer dtype numpy derives the spacing from the first two values
    cast to int, so the centres are `int(start) + i * (int(start + step) - int(start))`; the last window
    then often stops short of the far border and those voxels are never visited (0/0 = NaN ->
    class 0 in the mask, NaN in the one-hot map).  `window_centres` restates that rule explicitly
    (`np.int` is gone from numpy >= 1.24, where the reference raises AttributeError);
  * the averaged probabilities go through a second softmax before the argmax (trainer.py:93).

What runs where: every window is one forward of the model through the HIP kernels; the softmax +
accumulate and the divide + softmax + argmax + crop are two streaming HIP kernels
(`ru3d_predict_accumulate`, `ru3d_predict_merge`); the volumes `result` / `result_n` live in HBM as
[X, Y, Z, C] / [X, Y, Z] fp32 for the whole case and only the final mask crosses PCIe.
`patch_batch` windows are pushed through the network per forward (InstanceNorm is per sample, so the
result does not depend on it); the sums are still accumulated in the reference's window order.
"""
import ctypes
import math

import numpy as np
import torch

import _native as N
from _native import check, ptr, stream

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None


def padded_shape(shape, patch_size):
    """`pad(input, patch_size)` (transform.py:386-389): every axis at least as long as the patch."""
    return tuple(max(int(shape[i]), int(patch_size[i])) for i in range(3))


def pad_offset(orig, full):
    """Where `pad` puts the case inside the padded volume: crop_pad's centred box starts at
    (orig - full) // 2 <= 0 and that many voxels are padded in front (transform.py:401-432)."""
    return tuple(-((int(orig[i]) - int(full[i])) // 2) for i in range(3))


def crop_offset(orig, full):
    """Where the final `crop_pad(result, original_shape)` cuts (trainer.py:98): (full - orig) // 2.  For an
    odd difference this is one voxel less than `pad_offset` - the reference returns such a case shifted
    by one voxel along that axis, and so does this function."""
    return tuple((int(full[i]) - int(orig[i])) // 2 for i in range(3))


def window_centres(length, patch, step_per_patch):
    """Window centres along one axis of (padded) length `length` (trainer.py:29-40)."""
    start = patch // 2
    end = length - patch // 2
    num_steps = math.ceil((end - start) / (patch / step_per_patch))
    step = (end - start) / (num_steps + 1e-8)
    if step == 0:
        step = 9999999
    # np.arange(start, end + 1e-8, step, dtype=int): length from the float arguments, spacing from the
    # first two values truncated to int
    count = int(math.ceil((end + 1e-8 - start) / step))
    delta = int(start + step) - int(start)
    return [int(start) + i * delta for i in range(max(count, 0))]


def window_origins(shape, patch_size, step_per_patch):
    """Lower corners of all windows in the reference's loop order (x outer, z inner; trainer.py:55-67)."""
    axes = [window_centres(shape[i], patch_size[i], step_per_patch) for i in range(3)]
    return [(x - patch_size[0] // 2, y - patch_size[1] // 2, z - patch_size[2] // 2)
            for x in axes[0] for y in axes[1] for z in axes[2]], [len(a) for a in axes]


def predict_per_patch(input, model, num_classes=3, patch_size=(96, 96, 96), step_per_patch=4, verbose=True,
                      one_hot=False, patch_batch=1, return_device=False):
    """input: numpy (or torch, host or device) [X, Y, Z, C_in] (the reference's W,H,D,C case layout).  Returns the
    uint8 mask [X, Y, Z] (or the float32 [X, Y, Z, num_classes] probability map when one_hot) at the input's own shape,
    as a numpy array like the reference - or as the device tensor when return_device (predict_case keeps going on
    the GPU)."""
    device = next(model.parameters()).device
    N.require_device(next(model.parameters()), "model")
    patch_size = tuple(int(p) for p in patch_size)
    if any(p % 2 for p in patch_size):
        # the reference slices [c - p//2, c + p//2): an odd patch would feed the model p-1 voxels
        raise ValueError("predict_per_patch: patch_size must be even, got %s" % (patch_size,))
    if not torch.is_tensor(input):
        input = np.asarray(input)
    if input.ndim != 4:
        raise ValueError("predict_per_patch: expected a [X, Y, Z, C] volume, got shape %s" % (input.shape,))
    original_shape = tuple(int(s) for s in input.shape[:3])
    full = padded_shape(original_shape, patch_size)
    lo = pad_offset(original_shape, full)
    co = crop_offset(original_shape, full)
    cin = int(input.shape[3])

    # padded volume in HBM, NDHWC (= the case layout with a leading batch axis); zero padding as np.pad's default
    vol = torch.zeros((1,) + full + (cin,), dtype=torch.float32, device=device)
    vol[0, lo[0]:lo[0] + original_shape[0], lo[1]:lo[1] + original_shape[1], lo[2]:lo[2] + original_shape[2]] = \
        (input.to(device=device, dtype=torch.float32) if torch.is_tensor(input)
         else torch.from_numpy(np.ascontiguousarray(input, dtype=np.float32)).to(device))
    vol = vol.permute(0, 4, 1, 2, 3)                                   # [1, C, X, Y, Z] view

    origins, counts = window_origins(full, patch_size, step_per_patch)
    for (ox, oy, oz) in origins:
        if ox < 0 or oy < 0 or oz < 0 or ox + patch_size[0] > full[0] or oy + patch_size[1] > full[1] \
                or oz + patch_size[2] > full[2]:
            raise ValueError("predict_per_patch: window at %s leaves the %s volume" % ((ox, oy, oz), full))
    if verbose:
        print('Image Shape: {} Patch Size: {}'.format(full, patch_size))
        print('X step: %d Y step: %d Z step: %d' % tuple(counts))

    acc = torch.zeros(full + (num_classes,), dtype=torch.float32, device=device)
    cnt = torch.zeros(full, dtype=torch.float32, device=device)
    px, py, pz = patch_size
    patch_batch = max(1, int(patch_batch))
    bar = tqdm(total=len(origins)) if (verbose and tqdm is not None) else None
    model.eval()
    with torch.no_grad():
        for b0 in range(0, len(origins), patch_batch):
            group = origins[b0:b0 + patch_batch]
            x = N.new_act(len(group), cin, px, py, pz, torch.float32, device)
            for i, (ox, oy, oz) in enumerate(group):
                x[i].copy_(vol[0, :, ox:ox + px, oy:oy + py, oz:oz + pz])
            logits = model(x)
            if logits.shape[1] != num_classes:
                raise ValueError("predict_per_patch: model returns %d classes, num_classes is %d"
                                 % (logits.shape[1], num_classes))
            logits = N.to_ndhwc(logits)
            d = N.desc(logits)
            for i, (ox, oy, oz) in enumerate(group):
                check(N.lib.ru3d_predict_accumulate(ctypes.byref(d), N.dtype_code(logits.dtype), i, ptr(acc),
                                                    ptr(cnt), full[0], full[1], full[2], ox, oy, oz, stream()),
                      "predict_accumulate")
            if bar is not None:
                bar.update(len(group))
    if bar is not None:
        bar.close()
    if verbose:
        print('Merging all patchs...')
    sx, sy, sz = original_shape
    if one_hot:
        out = torch.empty(original_shape + (num_classes,), dtype=torch.float32, device=device)
    else:
        out = torch.empty(original_shape, dtype=torch.uint8, device=device)
    N.note_device(acc.device)
    check(N.lib.ru3d_predict_merge(ptr(acc), ptr(cnt), full[0], full[1], full[2], num_classes, co[0], co[1], co[2],
                                   sx, sy, sz, 1 if one_hot else 0, ptr(out), stream()), "predict_merge")
    return out if return_device else out.cpu().numpy()


# --------------------------------------------------------------------------- whole-case inference (trainer.py:101-133)
def _zoomed_shape(shape, scale):
    """Output shape of scipy.ndimage.zoom for an input shape and per-axis factors."""
    return tuple(int(round(s * z)) for s, z in zip(shape, scale))


def predict_case(case, model, target_spacing, normalize_stats, num_classes=3, patch_size=(96, 96, 96),
                 step_per_patch=4, verbose=True, one_hot=False, patch_batch=1):
    """reference trainer.py:101-133: resample the case to `target_spacing` and normalise it (data.py:222-283), run the
    sliding-window prediction, resize the prediction back to the case's shape.  Everything between the upload of the
    image and the download of the prediction runs on the device: the two resamplings are the order-1 zoom kernel of
    the augmentation path (label rule included), the sliding window is predict_per_patch."""
    import augment
    device = next(model.parameters()).device
    image = np.asarray(case['image'])
    if image.ndim == 3:
        image = image[..., None]
    orig_shape = tuple(int(s) for s in image.shape[:-1])
    affine = np.asarray(case['affine'], dtype=np.float64)
    stats = normalize_stats if isinstance(normalize_stats, list) else [normalize_stats]
    if verbose:
        print('Resampling the case for prediction...')
    spacing = np.array([np.linalg.norm(affine[i, :3]) for i in range(3)])
    scale = spacing / np.array(target_spacing, dtype=np.float64)
    vol = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(device)
    vol = augment.resample_image(vol, _zoomed_shape(orig_shape, scale))
    for c, s in enumerate(stats):                 # clip to the percentiles, then (x - mean) / (std + 1e-8)
        vol[..., c].clamp_(float(s['pct_00_5']), float(s['pct_99_5'])).sub_(float(s['mean'])).div_(float(s['std']) + 1e-8)
    vol = vol[..., :len(stats)]
    if verbose:
        print('Predicting the case...')
    pred = predict_per_patch(vol, model, num_classes, patch_size, step_per_patch, verbose, one_hot,
                             patch_batch=patch_batch, return_device=True)
    if verbose:
        print('Resizing the case to origial shape...')
    if one_hot:
        out = augment.resample_image(pred, orig_shape).cpu().numpy()
    else:
        out = augment.resample_label(pred, orig_shape).to(torch.uint8).cpu().numpy()
    case['pred'] = out
    case['affine'] = affine
    if verbose:
        print('All done!')
    return case
