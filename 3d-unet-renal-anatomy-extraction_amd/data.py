"""Drop-in `data` module: case files and the host-side case preparation of the reference (reference data.py).

`CaseDataset`, `load_case`, `save_case`, `save_pred`, `get_spacing`, `apply_scale`, `apply_translate`,
`resample_normalize_case`, `regions_crop_case` keep their names, arguments and case-dict layout
({'case_id', 'affine', 'image' float32 [X,Y,Z,C], 'label' int64 [X,Y,Z], 'pred'}).  NIfTI files go through nifti.py
(nibabel when it is installed, a numpy reader / writer of the NIfTI-1 subset the reference uses otherwise); the
affine decomposition restates transforms3d.affines.decompose / compose (a dependency of the reference that is absent
here) for the two helpers that use it.  `orient_crop_case` (data.py:117-172) reorients through this module's
restatement of nibabel's published orientation algebra (`io_orientation`, `apply_orientation`, `inv_ornt_aff`:
nibabel is a dependency of the reference that is absent here, so those three are pinned by their defining property -
every voxel keeps its world coordinate - not by nibabel outputs).
"""
from pathlib import Path

import numpy as np
import scipy.ndimage as ndi
import torch

import nifti
from transform import crop_pad_to_bbox, remove_small_region, rescale, split_dim  # noqa: F401
from utils import json_load, json_save  # noqa: F401

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, *a, **k):
        return x


def _case_id(path):
    return str(path).split('/')[-1].split('.')[0]


def _with_channel(image):
    """Cases carry a trailing channel axis; single-modality files are stored as plain 3-D volumes."""
    return image[..., None] if image.ndim == 3 else image


class CaseDataset(torch.utils.data.Dataset):
    """Folder of `<id>.image.nii.gz` (+ `<id>.label.nii.gz`) files -> case dicts (data.py:14-52)."""

    def __init__(self, load_dir, transform=None, load_meta=False):
        super().__init__()
        self.load_dir = Path(load_dir)
        self.transform = transform
        self.image_files = sorted(self.load_dir.glob('*.image.nii.gz'))
        self.label_files = sorted(self.load_dir.glob('*.label.nii.gz'))
        self.load_label = len(self.image_files) == len(self.label_files) and len(self.label_files) > 0

    def __getitem__(self, index):
        case = load_case(self.image_files[index], self.label_files[index] if self.load_label else None)
        return self.transform(case) if self.transform else case

    def __len__(self):
        return len(self.image_files)


def load_case(image_file, label_file=None):
    image, affine, _ = nifti.load(image_file)
    case = {'case_id': _case_id(image_file), 'affine': affine, 'image': _with_channel(image.astype(np.float32))}
    if label_file:
        label, _, _ = nifti.load(label_file)
        case['label'] = label.astype(np.int64)
    return case


def save_case(case, save_dir):
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    nifti.save(case['image'].astype(np.float32), case['affine'], save_dir / ('%s.image.nii.gz' % case['case_id']))
    for key in ('label', 'pred'):
        if key in case:
            nifti.save(case[key].astype(np.uint8), case['affine'], save_dir / ('%s.%s.nii.gz' % (case['case_id'], key)))


def save_pred(case, save_dir):
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    nifti.save(case['pred'].astype(np.uint8), case['affine'], save_dir / ('%s.pred.nii.gz' % case['case_id']))


# ------------------------------------------------------------------ affine helpers (data.py:55-76)
def get_spacing(affine):
    return tuple(float(np.linalg.norm(affine[i, :3])) for i in range(3))


def _decompose(affine):
    """transforms3d.affines.decompose: A = T . R . diag(Z) . S with S upper-triangular unit shears."""
    a = np.asarray(affine, dtype=np.float64)
    t = a[:3, 3].copy()
    rzs = a[:3, :3]
    zs = np.linalg.cholesky(rzs.T @ rzs).T
    z = np.diag(zs).copy()
    shears = zs / z[:, None]
    r = rzs @ np.linalg.inv(zs)
    if np.linalg.det(r) < 0:
        z[0] *= -1
        zs[0] *= -1
        r = rzs @ np.linalg.inv(zs)
    return t, r, z, shears


def _compose(t, r, z, shears):
    a = np.eye(4)
    a[:3, :3] = r @ np.diag(z) @ shears
    a[:3, 3] = t
    return a


def apply_scale(affine, scale):
    t, r, z, s = _decompose(affine)
    return _compose(t, r, z * np.array(scale), s)


def apply_translate(affine, offset):
    t, r, z, s = _decompose(affine)
    return _compose(t + np.array(offset), r, z, s)


# ------------------------------------------------------------------ orientation (nibabel.orientations, restated)
def io_orientation(affine, tol=None):
    """Orientation of the voxel axes closest to the world axes: [[output axis, +1 | -1 flip], ...] per input axis
    (nibabel.orientations.io_orientation: polar part of the direction cosines by SVD, then a greedy
    largest-component assignment that uses every output axis once)."""
    affine = np.asarray(affine, dtype=np.float64)
    q, p = affine.shape[0] - 1, affine.shape[1] - 1
    rzs = affine[:q, :p]
    zooms = np.sqrt(np.sum(rzs * rzs, axis=0))
    zooms[zooms == 0] = 1
    rs = rzs / zooms
    u, sv, vt = np.linalg.svd(rs, full_matrices=False)
    if tol is None:
        tol = sv.max() * max(rs.shape) * np.finfo(sv.dtype).eps
    keep = sv > tol
    r = np.dot(u[:, keep], vt[keep])
    ornt = np.ones((p, 2), dtype=np.float64) * np.nan
    for in_ax in range(p):
        col = r[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            r[out_ax, :] = 0          # this output axis is taken
    return ornt


def apply_orientation(arr, ornt):
    """Flip, then permute, the leading axes of `arr` as `ornt` says (nibabel.orientations.apply_orientation)."""
    t = np.asarray(arr)
    ornt = np.asarray(ornt)
    n = ornt.shape[0]
    if t.ndim < n:
        raise ValueError("data array has fewer dimensions than the orientation")
    if np.any(np.isnan(ornt[:, 0])):
        raise ValueError("cannot reorient along a dropped axis")
    for ax, flip in enumerate(ornt[:, 1]):
        if flip == -1:
            t = np.flip(t, axis=ax)
    full = np.arange(t.ndim)
    full[:n] = np.argsort(ornt[:, 0])
    return t.transpose(full)


def inv_ornt_aff(ornt, shape):
    """Affine from the reoriented array's voxel indices back to the original's (nibabel.orientations.inv_ornt_aff)."""
    ornt = np.asarray(ornt)
    p = ornt.shape[0]
    shape = np.array(shape)[:p]
    undo_reorder = np.eye(p + 1)[list(ornt[:, 0].astype(int)) + [p], :]
    undo_flip = np.diag(list(ornt[:, 1]) + [1.0])
    center = -(shape - 1) / 2.0
    undo_flip[:p, p] = ornt[:, 1] * center - center
    return np.dot(undo_flip, undo_reorder)


def reorient(array, affine, ornt):
    """(array, affine) after `ornt` - what nibabel's `Nifti1Pair(array, affine).as_reoriented(ornt)` holds."""
    return apply_orientation(array, ornt), np.dot(np.asarray(affine, dtype=np.float64), inv_ornt_aff(ornt, np.asarray(array).shape))


def orient_crop_case(case, air=-200):
    """data.py:117-172: reorient the case to the closest-to-canonical axes, then crop it to the bounding box of the voxels
    above `air` (in any channel); 'bbox' records the box, the affine moves with the crop."""
    case = case.copy()
    ornt = io_orientation(case['affine'])
    image, new_affine = reorient(case['image'], case['affine'], ornt)
    image = image.astype(np.float32)
    if 'label' in case:
        label = apply_orientation(case['label'], ornt).astype(np.int64)
    if image.ndim == 3:
        image = image[..., None]
    lo, hi = [], []
    for channel in split_dim(image):
        pos = np.array(np.where(channel > air))
        lo.append(pos.min(axis=1))
        hi.append(pos.max(axis=1))
    bbox = np.array([np.array(lo).min(axis=0), np.array(hi).max(axis=0)]).T          # (3, 2): as in the reference, the
    bbox_c = np.concatenate([bbox, [[0, image.shape[-1]]]])                          # upper bound is the last index itself
    case['image'] = crop_pad_to_bbox(image, bbox_c)
    case['bbox'] = bbox
    if 'label' in case:
        case['label'] = crop_pad_to_bbox(label, bbox)
    case['affine'] = apply_translate(new_affine, bbox[:, 0] * np.array(get_spacing(new_affine)))
    return case


# ------------------------------------------------------------------ preparation (data.py:222-283, 464-492)
def resample_normalize_case(case, target_spacing, normalize_stats):
    """Resample image (and label) to `target_spacing`, clip every channel to its [pct_00_5, pct_99_5] and normalise it
    with (x - mean) / (std + 1e-8).  Host version (scipy zoom); trainer.predict_case runs the same arithmetic on the
    device."""
    case = case.copy()
    stats = normalize_stats if isinstance(normalize_stats, list) else [normalize_stats]
    scale = np.array(get_spacing(case['affine'])) / np.array(target_spacing)
    image = rescale(case['image'], scale, multi_class=True)
    channels = []
    for c, s in enumerate(stats):
        clipped = np.clip(image[..., c], s['pct_00_5'], s['pct_99_5'])
        channels.append((clipped - s['mean']) / (s['std'] + 1e-8))
    case['image'] = np.stack(channels, axis=-1)
    if 'label' in case:
        case['label'] = rescale(case['label'], scale, is_label=True)
    case['affine'] = apply_scale(case['affine'], 1 / scale)
    return case


def regions_crop_case(case, threshold=0, padding=20, based_on='label'):
    """Connected foreground regions of the label (or prediction), each cropped with `padding` millimetres around it."""
    based = remove_small_region(np.array(case[based_on] > 0), threshold)
    labels, _ = ndi.label(based)
    spacing = np.array(get_spacing(case['affine']))
    pad_vox = np.round(padding / spacing).astype(int)
    regions = []
    for i, sl in enumerate(ndi.find_objects(labels)):
        bbox = np.array([[sl[d].start - pad_vox[d], sl[d].stop + pad_vox[d]] for d in range(3)])
        bbox_c = np.concatenate([bbox, [[0, case['image'].shape[-1]]]])
        region = {'case_id': '%s_%03d' % (case['case_id'], i),
                  'affine': apply_translate(case['affine'], bbox[:, 0] * spacing),
                  'bbox': bbox,
                  'image': crop_pad_to_bbox(case['image'], bbox_c)}
        if 'label' in case:
            region['label'] = crop_pad_to_bbox(case['label'], bbox)
        regions.append(region)
    return regions
