"""On-device patch sampling / augmentation against the reference's own transform pipeline
(fixture G7, produced by tests/golden/make_golden_augment.py from the reference's transform.py) and against the numpy
oracle on larger cases.  Tolerances: labels identical; resampled image 2e-6 abs (float64 trilinear interpolation
rounded to float32 on both sides); after the intensity chain 2e-5 abs (float32 mean and powf differ in the last
ulps between numpy and the device).  Run with `-m gpu`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import augment  # noqa: E402
from oracle import augment_oracle as A  # noqa: E402

DEV = torch.device("cuda:0")

G7_CASES = {
    "iia_like": dict(scale=0.1, crop_mode="random"),
    "iia_like_b": dict(scale=0.1, crop_mode="random"),
    "binary_label": dict(scale=0.2, crop_mode="random"),
    "pads": dict(scale=0.1, crop_mode="random"),
    "center_two_ch": dict(scale=[0.8, 1.3], crop_mode="center"),
    "margin_enforce": dict(scale=0.1, crop_mode="random", crop_margin=4, enforce_label_indices=[2]),
}


@pytest.mark.parametrize("tag", sorted(G7_CASES))
def test_g7_device_pipeline_matches_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "g7_augment.npz"))
    patch = tuple(int(v) for v in z[tag + "/patch"])
    case = augment.DeviceCase(z[tag + "/image_in"], z[tag + "/label_in"], DEV)
    np.random.seed(int(z[tag + "/seed"]))
    aug = augment.DeviceAugment(crop_size=patch, **G7_CASES[tag])
    img, lab = aug.sample(case)
    torch.cuda.synchronize()
    assert tuple(img.shape) == z[tag + "/image_out"].shape and img.dtype == torch.float32
    assert lab.dtype == torch.int64
    assert np.array_equal(lab.cpu().numpy(), z[tag + "/label_out"].astype(np.int64))
    assert np.abs(img.cpu().numpy() - z[tag + "/image_out"]).max() <= 2e-5
    # the stage before the intensity chain (resample + mirror) on its own
    np.random.seed(int(z[tag + "/seed"]))
    plain = augment.DeviceAugment(crop_size=patch, contrast=None, brightness=None, gamma=None, **G7_CASES[tag])
    img2, lab2 = plain.sample(case)
    want = np.moveaxis(z[tag + "/after_mirror"], -1, 0)
    assert np.abs(img2.cpu().numpy() - want).max() <= 2e-6
    assert torch.equal(lab, lab2)


@pytest.mark.parametrize("classes,dtype", [(4, np.uint8), (2, np.int64), (3, np.uint8)])
def test_training_size_patch_vs_numpy_oracle(classes, dtype):
    """A 128^3 patch (the reference's patch_size, nb_train_iia.py:29) from a 176 x 160 x 144 case, same numpy seed on
    both sides: labels identical, image within 2e-5."""
    rng = np.random.RandomState(7 + classes)
    shape = (176, 160, 144)
    g = np.meshgrid(*[np.linspace(-1, 1, s, dtype=np.float32) for s in shape], indexing="ij")
    img = (np.sin(4 * g[0]) * np.cos(3 * g[1]) + g[2] ** 2 + 0.05 * rng.randn(*shape)).astype(np.float32)[..., None]
    lab = np.zeros(shape, dtype=dtype)
    r = np.sqrt(g[0] ** 2 + (1.2 * g[1]) ** 2 + g[2] ** 2)
    lab[r < 0.8] = 1
    if classes > 2:
        lab[np.sqrt((g[0] - 0.2) ** 2 + g[1] ** 2 + g[2] ** 2) < 0.35] = 2
    if classes > 3:
        lab[np.sqrt((g[0] + 0.3) ** 2 + (g[1] - 0.1) ** 2 + g[2] ** 2) < 0.2] = 3
    patch = (128, 128, 128)
    np.random.seed(42)
    want_img, want_lab, _, _ = A.pipeline(img, lab, patch, scale=0.1, crop_mode="random")
    case = augment.DeviceCase(img, lab, DEV)
    np.random.seed(42)
    got_img, got_lab = augment.DeviceAugment(scale=0.1, crop_size=patch, crop_mode="random").sample(case)
    torch.cuda.synchronize()
    assert np.array_equal(got_lab.cpu().numpy(), want_lab.astype(np.int64))
    assert np.abs(got_img.cpu().numpy() - want_img).max() <= 2e-5


def test_batch_feeds_the_model_and_call_interface():
    import loss as L
    import network
    rng = np.random.RandomState(1)
    cases = []
    for i in range(2):
        img = rng.randn(40, 40, 40, 1).astype(np.float32)
        lab = (rng.rand(40, 40, 40) > 0.7).astype(np.uint8)
        cases.append(augment.DeviceCase(img, lab, DEV))
    np.random.seed(3)
    aug = augment.DeviceAugment(scale=0.1, crop_size=32, crop_mode="random")
    b = aug.batch(cases, 3)
    assert tuple(b["image"].shape) == (3, 1, 32, 32, 32) and tuple(b["label"].shape) == (3, 32, 32, 32)
    assert b["image"].is_cuda and b["label"].dtype == torch.int64 and int(b["label"].max()) <= 1
    torch.manual_seed(0)
    model = network.ResUnet3D(2, 8, 1, 2).to(DEV)
    loss = L.HybirdLoss()(model(b["image"]), b["label"])
    loss.backward()
    assert torch.isfinite(loss)
    out = aug({"image": rng.randn(36, 36, 36, 1).astype(np.float32), "label": np.zeros((36, 36, 36), np.uint8),
               "case_id": "x"})
    assert out["case_id"] == "x" and tuple(out["image"].shape) == (1, 32, 32, 32) and out["image"].is_cuda
