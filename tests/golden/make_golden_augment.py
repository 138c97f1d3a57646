#!/usr/bin/env python3
"""Golden fixtures of the patch sampling / augmentation path from the *reference itself*.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's own `transform.py` (numpy + scipy
only) and runs the training pipeline of nb_train_iia.py:30-39 -

    RandomRescaleCrop(0.1, patch, crop_mode='random') -> RandomMirror((.5,.5,.5)) -> RandomContrast(0.1)
    -> RandomBrightness(0.1) -> RandomGamma(0.1) -> ToTensor()

- on small synthetic cases under np.random.seed(s), freezing inputs, the seed and the outputs into
tests/golden/g7_augment.npz.  Only arrays are stored.  `RandomRescaleCrop.__call__` uses `np.int` (removed from numpy
in 1.24: the reference crashes on this container's numpy 2.2 at transform.py:617); the alias is restored for the call.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_augment.py
"""
import importlib.util
import os
import sys

import numpy as np

REF = os.environ.get("RU3D_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_case(shape, channels, classes, seed):
    """Smooth-ish image (so interpolation matters) + blocky label with `classes` classes."""
    rng = np.random.RandomState(seed)
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij")
    img = np.stack([np.sin(3 * x + c) * np.cos(2 * y) + 0.5 * z ** 2 + 0.1 * rng.randn(*shape) for c in range(channels)],
                   axis=-1).astype(np.float32)
    lab = np.zeros(shape, dtype=np.uint8)
    r = np.sqrt((x * 1.1) ** 2 + y ** 2 + (z * 0.9) ** 2)
    lab[r < 0.75] = 1
    if classes > 2:
        lab[np.sqrt((x - 0.2) ** 2 + (y + 0.1) ** 2 + z ** 2) < 0.3] = 2
    if classes > 3:
        lab[np.sqrt((x + 0.3) ** 2 + (y - 0.2) ** 2 + (z - 0.1) ** 2) < 0.2] = 3
    return img, lab


CASES = [
    # tag, volume shape, channels, classes, patch, kwargs of RandomRescaleCrop, seed
    ("iia_like", (40, 36, 28), 1, 4, (24, 24, 16), dict(scale=0.1, crop_mode="random"), 1),
    ("iia_like_b", (40, 36, 28), 1, 4, (24, 24, 16), dict(scale=0.1, crop_mode="random"), 2),
    ("binary_label", (30, 30, 30), 1, 2, (16, 16, 16), dict(scale=0.2, crop_mode="random"), 3),
    ("pads", (20, 18, 14), 1, 3, (24, 20, 16), dict(scale=0.1, crop_mode="random"), 4),          # crop > volume: pad
    ("center_two_ch", (26, 24, 22), 2, 3, (16, 12, 10), dict(scale=[0.8, 1.3], crop_mode="center"), 5),
    ("margin_enforce", (48, 40, 32), 1, 3, (16, 16, 16),
     dict(scale=0.1, crop_mode="random", crop_margin=4, enforce_label_indices=[2]), 6),
]


def main():
    T = _load("transform")
    out = {}
    had_int = hasattr(np, "int")
    if not had_int:
        np.int = int
    try:
        for tag, shape, ch, classes, patch, kw, seed in CASES:
            img, lab = make_case(shape, ch, classes, 100 + seed)
            steps = [T.RandomRescaleCrop(kw["scale"], patch, **{k: v for k, v in kw.items() if k != "scale"}),
                     T.RandomMirror((0.5, 0.5, 0.5)), T.RandomContrast(0.1), T.RandomBrightness(0.1),
                     T.RandomGamma(0.1), T.ToTensor()]
            np.random.seed(seed)
            case = {"image": img.copy(), "label": lab.copy()}
            stages = {}
            for st in steps:
                case = st(case)
                stages[type(st).__name__] = np.ascontiguousarray(case["image"]).copy()
            out[tag + "/image_in"] = img
            out[tag + "/label_in"] = lab
            out[tag + "/seed"] = np.int64(seed)
            out[tag + "/patch"] = np.array(patch, dtype=np.int64)
            out[tag + "/image_out"] = np.ascontiguousarray(case["image"]).astype(np.float32)      # [C, x, y, z]
            out[tag + "/label_out"] = np.ascontiguousarray(case["label"])
            out[tag + "/after_crop"] = stages["RandomRescaleCrop"].astype(np.float32)            # [x, y, z, C]
            out[tag + "/after_mirror"] = stages["RandomMirror"].astype(np.float32)
            print(tag, case["image"].shape, case["label"].shape, case["label"].dtype, np.unique(case["label"]))
    finally:
        if not had_int:
            del np.int
    np.savez_compressed(os.path.join(OUT, "g7_augment.npz"), **out)
    print("wrote g7_augment.npz")


if __name__ == "__main__":
    main()
