"""Host-side data path (CPU): the drop-in `transform`, `data`, `nifti`, `utils` modules.

  * transform.py's CPU classes reproduce the reference's pipeline on the G7 fixtures (same numpy seed -> same patch);
  * nifti.py round-trips arrays + affines through .nii.gz and reads a hand-assembled big-endian / scaled header;
  * data.py: CaseDataset over a folder of case files, affine helpers, resample_normalize_case, regions_crop_case."""
import gzip
import os
import struct

import numpy as np
import pytest

import data
import nifti
import transform as T

G7_CASES = {
    "iia_like": dict(scale=0.1, crop_mode="random"),
    "binary_label": dict(scale=0.2, crop_mode="random"),
    "pads": dict(scale=0.1, crop_mode="random"),
    "center_two_ch": dict(scale=[0.8, 1.3], crop_mode="center"),
    "margin_enforce": dict(scale=0.1, crop_mode="random", crop_margin=4, enforce_label_indices=[2]),
}


@pytest.mark.parametrize("tag", sorted(G7_CASES))
def test_cpu_transform_classes_reproduce_the_reference_pipeline(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "g7_augment.npz"))
    patch = tuple(int(v) for v in z[tag + "/patch"])
    kw = dict(G7_CASES[tag])
    pipe = T.Compose([T.RandomRescaleCrop(kw.pop("scale"), patch, **kw), T.RandomMirror((0.5, 0.5, 0.5)),
                      T.RandomContrast(0.1), T.RandomBrightness(0.1), T.RandomGamma(0.1), T.ToTensor()])
    np.random.seed(int(z[tag + "/seed"]))
    out = pipe({"image": z[tag + "/image_in"].copy(), "label": z[tag + "/label_in"].copy()})
    assert np.array_equal(out["label"], z[tag + "/label_out"])
    assert out["image"].dtype == np.float32 and out["image"].shape == z[tag + "/image_out"].shape
    assert np.abs(out["image"] - z[tag + "/image_out"]).max() <= 1e-6      # same scipy, same float32 arithmetic


def test_transform_helpers():
    lab = np.array([[0, 1, 2], [3, 2, 1]], dtype=np.uint8)
    assert np.array_equal(T.combination_labels(lab, [1, 2], 4), np.array([[0, 1, 1], [2, 1, 1]], dtype=np.uint8))
    assert np.array_equal(T.combination_labels(lab, [[0, 3], [1, 2]], 4), np.array([[0, 1, 1], [0, 1, 1]]))
    oh = T.to_one_hot(lab, 4, to_tensor=True)
    assert oh.shape == (4, 2, 3) and oh.dtype == np.uint8 and oh.sum(axis=0).min() == 1
    x = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    assert T.to_numpy(T.to_tensor(x)).shape == x.shape
    assert T.pad(x, (4, 3, 6)).shape == (4, 3, 6) and T.crop_pad(x, (1, 2, 2)).shape == (1, 2, 2)
    assert T.CenterCrop(2)({"image": np.zeros((4, 4, 4, 1), np.float32), "label": np.zeros((4, 4, 4), np.uint8)})[
        "image"].shape == (2, 2, 2, 1)
    blob = np.zeros((8, 8), dtype=np.uint8)
    blob[0, 0] = 1
    blob[4:7, 4:7] = 1
    assert T.remove_small_region(blob.copy(), 4).sum() == 9


def test_nifti_roundtrip_and_foreign_header(tmp_path):
    rng = np.random.RandomState(0)
    aff = np.array([[0.8, 0.05, 0, -10], [0, 0.75, 0.1, 5], [0, 0, 3.0, 2], [0, 0, 0, 1]])
    for arr in (rng.randn(5, 6, 7).astype(np.float32), (rng.rand(4, 5, 6, 2) * 100).astype(np.int16),
                (rng.rand(3, 4, 5) * 3).astype(np.uint8)):
        p = str(tmp_path / "a.nii.gz")
        nifti.save(arr, aff, p)
        back, aff2, _ = nifti.load(p)
        assert back.shape == arr.shape and np.array_equal(back, arr.astype(np.float64))
        assert np.allclose(aff2, aff, atol=1e-6)
    # a big-endian int16 file with scl_slope / scl_inter and only a qform (what some scanners write)
    hdr = bytearray(352)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">hh", hdr, 70, 4, 16)
    struct.pack_into(">8f", hdr, 76, 1.0, 2.0, 2.0, 5.0, 1, 1, 1, 1)
    struct.pack_into(">fff", hdr, 108, 352.0, 0.5, 10.0)
    struct.pack_into(">hh", hdr, 252, 1, 0)
    struct.pack_into(">6f", hdr, 256, 0.0, 0.0, 0.0, 1.0, 2.0, 3.0)
    hdr[344:348] = b"n+1\x00"
    vox = np.arange(24, dtype=">i2").reshape((2, 3, 4), order="F")
    p = str(tmp_path / "be.nii.gz")
    with gzip.open(p, "wb") as f:
        f.write(bytes(hdr) + vox.tobytes(order="F"))
    arr, aff3, _ = nifti.load(p)
    assert np.array_equal(arr, vox.astype(np.float64) * 0.5 + 10.0)
    assert np.allclose(aff3, np.array([[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 5, 3], [0, 0, 0, 1.0]]))
    with pytest.raises(ValueError):
        nifti.load(__file__)


def test_case_dataset_and_preparation(tmp_path):
    rng = np.random.RandomState(1)
    aff = np.diag([1.0, 1.0, 2.5, 1.0])
    for i in range(2):
        img = (rng.randn(12, 10, 8) * 50 + 100).astype(np.float32)
        lab = np.zeros((12, 10, 8), dtype=np.uint8)
        lab[2:6, 2:6, 1:4] = 1
        lab[8:11, 6:9, 5:7] = 2
        data.save_case({"case_id": "case_%02d" % i, "affine": aff, "image": img[..., None], "label": lab}, tmp_path)
    ds = data.CaseDataset(tmp_path)
    assert len(ds) == 2 and ds.load_label
    case = ds[1]
    assert case["case_id"] == "case_01" and case["image"].shape == (12, 10, 8, 1) and case["image"].dtype == np.float32
    assert case["label"].dtype == np.int64 and set(np.unique(case["label"])) == {0, 1, 2}
    assert np.allclose(data.get_spacing(case["affine"]), (1.0, 1.0, 2.5))
    assert np.allclose(data.get_spacing(data.apply_scale(case["affine"], (2, 2, 0.4))), (2.0, 2.0, 1.0))
    assert np.allclose(data.apply_translate(case["affine"], (3, 4, 5))[:3, 3], (3, 4, 5))
    sheared = np.array([[1, 0.2, 0, 0], [0, 1, 0.1, 0], [0, 0, 2, 0], [0, 0, 0, 1.0]])
    assert np.allclose(data._compose(*data._decompose(sheared)), sheared)
    stats = {"mean": 100.0, "std": 50.0, "pct_00_5": 0.0, "pct_99_5": 200.0}
    rs = data.resample_normalize_case(case, (2.0, 2.0, 2.5), stats)
    assert rs["image"].shape == (6, 5, 8, 1) and rs["label"].shape == (6, 5, 8)
    assert np.allclose(data.get_spacing(rs["affine"]), (2.0, 2.0, 2.5))
    assert rs["image"].min() >= -2.0 - 1e-6 and rs["image"].max() <= 2.0 + 1e-6
    regions = data.regions_crop_case(case, threshold=0, padding=2)
    assert len(regions) == 2 and regions[0]["case_id"] == "case_01_000"
    assert regions[0]["image"].shape[:3] == regions[0]["label"].shape and regions[0]["bbox"].shape == (3, 2)
    case["pred"] = case["label"].astype(np.uint8)
    data.save_pred(case, tmp_path / "pred")
    import trainer
    assert trainer.evaluate(tmp_path / "case_01.label.nii.gz", tmp_path / "pred" / "case_01.pred.nii.gz") == [1.0, 1.0]


# --------------------------------------------------------------------------- cascade / evaluation glue (fixture G9)
def _g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_cascade.npz"))


def test_evaluate_case_and_batch_evaluate_vs_reference_fixture(golden_dir, tmp_path, capsys):
    """trainer.evaluate_case (reference trainer.py:348-356) on the G9 label / prediction pair = the reference's own
    numbers; evaluate / batch_evaluate (:359-400) read the same pair back from .nii.gz files."""
    import trainer
    z = _g9(golden_dir)
    got = trainer.evaluate_case({"label": z["eval_label"], "pred": z["pred"]})
    assert len(got) == int(z["eval_label"].max()) == 3
    assert np.allclose(got, z["eval_dice"], rtol=0, atol=1e-6)
    ldir, pdir = tmp_path / "labels", tmp_path / "preds"
    ldir.mkdir()
    pdir.mkdir()
    aff = np.diag([1.6, 1.6, 3.0, 1.0])
    for i in range(2):      # case 1: a perfect prediction
        nifti.save(z["eval_label"], aff, ldir / ("case_%d.nii.gz" % i))
        nifti.save(z["pred"] if i == 0 else z["eval_label"], aff, pdir / ("case_%d.nii.gz" % i))
    assert np.allclose(trainer.evaluate(ldir / "case_0.nii.gz", pdir / "case_0.nii.gz"), z["eval_dice"], atol=1e-6)
    res = trainer.batch_evaluate(ldir, pdir)
    assert len(res) == 2 and np.allclose(res[0], z["eval_dice"], atol=1e-6) and np.allclose(res[1], 1.0, atol=1e-6)
    assert "label_1:" in capsys.readouterr().out
    assert len(trainer.batch_evaluate(ldir, pdir, data_range=[1])) == 1


def test_regions_crop_case_vs_reference_fixture(golden_dir):
    """data.regions_crop_case on the reference's coarse mask gives the reference's regions (bounding boxes incl. the
    millimetre padding) and crops that carry the padded box's shape."""
    z = _g9(golden_dir)
    case = {"case_id": "g9", "image": z["image"], "affine": z["affine"], "pred": z["coarse_pred"]}
    thr, pad = int(z["scalars"][1]), int(z["scalars"][2])
    regions = data.regions_crop_case(case, thr, pad, "pred")
    assert len(regions) == len(z["regions"]) == 2
    for r, bbox in zip(regions, z["regions"]):
        assert np.array_equal(r["bbox"], bbox)
        assert r["image"].shape[:3] == tuple(int(b[1] - b[0]) for b in bbox)
        # the affine moves with the box: its origin is the box's first voxel in world coordinates
        assert np.allclose(r["affine"][:3, 3], z["affine"][:3, 3] + bbox[:, 0] * np.array(data.get_spacing(z["affine"])))


def test_orientation_algebra_keeps_world_coordinates():
    """data.io_orientation / apply_orientation / inv_ornt_aff (nibabel's published algorithm, restated - nibabel is not
    installed here, so the pin is the defining property): after reorientation the axes are the closest to canonical
    (diagonally dominant, positive) and every voxel keeps its value and its world coordinate."""
    rng = np.random.RandomState(0)
    for trial in range(40):
        perm, flips = rng.permutation(3), rng.choice([-1, 1], 3)
        r = np.zeros((3, 3))
        for i in range(3):
            r[perm[i], i] = flips[i]
        a, b, c = rng.randn(3) * 0.12        # a small rotation on top of the signed permutation
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        aff = np.eye(4)
        aff[:3, :3] = rx @ ry @ rz @ r @ np.diag(rng.rand(3) * 2 + 0.5)
        aff[:3, 3] = rng.randn(3) * 30
        shape = tuple(rng.randint(3, 8, 3))
        arr = rng.rand(*shape)
        ornt = data.io_orientation(aff)
        assert sorted(ornt[:, 0]) == [0, 1, 2] and set(np.abs(ornt[:, 1])) == {1}
        arr2, aff2 = data.reorient(arr, aff, ornt)
        assert all(np.argmax(np.abs(aff2[:3, i])) == i and aff2[i, i] > 0 for i in range(3))
        idx = np.array([rng.randint(0, s) for s in shape])
        j = np.linalg.solve(data.inv_ornt_aff(ornt, shape), np.append(idx, 1))[:3]
        jj = np.round(j).astype(int)
        assert np.allclose(j, jj) and arr2[tuple(jj)] == arr[tuple(idx)]
        assert np.allclose(aff2 @ np.append(jj, 1), aff @ np.append(idx, 1))
    # the identity orientation changes nothing
    ornt = data.io_orientation(np.diag([2.0, 1.0, 3.0, 1.0]))
    assert np.array_equal(ornt, [[0, 1], [1, 1], [2, 1]])


def test_orient_crop_case_boxes_the_non_air_voxels():
    """data.orient_crop_case (reference data.py:117-172) on a flipped / permuted acquisition: canonical axes, the box of
    the voxels above `air` (upper bound = last index, as the reference computes it), label cropped alike, affine moved."""
    rng = np.random.RandomState(3)
    vol = np.full((12, 10, 9), -1000.0, dtype=np.float32)
    vol[3:8, 2:7, 4:8] = rng.rand(5, 5, 4).astype(np.float32) * 100
    label = (vol > 0).astype(np.int64)
    aff = np.array([[0, -1.5, 0, 10.0], [2.0, 0, 0, -4.0], [0, 0, -3.0, 7.0], [0, 0, 0, 1.0]])   # axes swapped, two flipped
    case = data.orient_crop_case({"case_id": "c", "image": vol, "label": label, "affine": aff}, air=-200)
    ornt = data.io_orientation(aff)
    vol_r, aff_r = data.reorient(vol, aff, ornt)
    pos = np.array(np.where(vol_r > -200))
    bbox = np.array([pos.min(axis=1), pos.max(axis=1)]).T
    assert np.array_equal(case["bbox"], bbox)
    assert case["image"].shape == tuple(bbox[:, 1] - bbox[:, 0]) + (1,)
    assert np.array_equal(case["image"][..., 0], vol_r[tuple(slice(b[0], b[1]) for b in bbox)])
    assert np.array_equal(case["label"], data.apply_orientation(label, ornt)[tuple(slice(b[0], b[1]) for b in bbox)])
    assert np.allclose(case["affine"][:3, 3], (aff_r @ np.append(bbox[:, 0], 1))[:3])
    assert np.allclose(data.get_spacing(case["affine"]), data.get_spacing(aff_r))
