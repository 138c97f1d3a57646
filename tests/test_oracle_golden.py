"""Pins the CPU oracle (oracle/unet_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _sub(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_g1_forward_bitexact(golden_dir):
    z = _load(golden_dir, "g1_config1.npz")
    w = _sub(z, "w/")
    x = torch.from_numpy(z["x"])
    logits = O.unet_forward(x, w, num_pool=2)
    ref = torch.from_numpy(z["logits"])
    assert torch.equal(logits.argmax(1).to(torch.uint8), torch.from_numpy(z["argmax"]))
    # same conv primitive; InstanceNorm restated as mean/var ops -> a few ulp of drift
    assert (logits - ref).abs().max().item() <= 2e-5


def test_g1_losses(golden_dir):
    z = _load(golden_dir, "g1_config1.npz")
    logits = torch.from_numpy(z["logits"])
    y = torch.from_numpy(z["y"].astype(np.int64))
    vals = {
        "hybird": O.hybird_loss(logits, y),
        "diceloss": O.dice_loss(logits, y),
        "focal": O.focal_loss(logits, y),
        "dice": O.dice_metric(logits, y),
    }
    for k, v in vals.items():
        assert abs(v.item() - float(z["loss/" + k])) <= 2e-7 * max(1.0, abs(float(z["loss/" + k]))), k


def _adam_close(a, b, key, steps, lr=1e-4):
    """Adam turns a gradient into a step of size ~lr whatever its magnitude, so an element whose
    true gradient is ~0 (every conv bias that feeds InstanceNorm; a few weight elements) moves by
    +-lr on rounding noise.  Such elements are only bounded by 2*lr*steps; the bulk must agree tightly."""
    d = (a - b).abs()
    assert d.max().item() <= 2.1 * lr * steps, key
    if key.endswith(("conv1.bias", "conv2.bias", "up.0.bias")):
        return
    frac_loose = (d > 2e-6 * steps).float().mean().item()
    assert frac_loose <= 0.01, (key, frac_loose)


def test_g1_grads_and_adam(golden_dir):
    z = _load(golden_dir, "g1_config1.npz")
    w = _sub(z, "w/")
    x = torch.from_numpy(z["x"])
    y = torch.from_numpy(z["y"].astype(np.int64))
    loss, logits, grads = O.train_step(w, x, y, 2)
    gref = _sub(z, "g/")
    assert set(grads) == set(gref)
    none_keys = set(z["none_grad_keys"].tolist())
    assert none_keys == set(O.unused_param_keys(w)) == set(w) - set(grads) - {k for k in w if "num_batches" in k}
    for k in gref:
        a, b = grads[k], gref[k]
        assert (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), k
    # 3 Adam steps
    state = {}
    wcur = {k: v.clone() for k, v in w.items()}
    losses = []
    for step in range(3):
        l, _, g = O.train_step(wcur, x, y, 2)
        wcur = O.adam_step(wcur, g, state)
        losses.append(l.item())
        if step == 0:
            ref1 = _sub(z, "adam1/")
            for k in ref1:
                _adam_close(wcur[k], ref1[k], k, 1)
    ref3 = _sub(z, "adam3/")
    for k in ref3:
        _adam_close(wcur[k], ref3[k], k, 3)
    assert np.allclose(losses, z["adam_losses"], rtol=0, atol=2e-6)


def test_g2_dropout_masks(golden_dir):
    z1 = _load(golden_dir, "g1_config1.npz")
    z = _load(golden_dir, "g2_dropout.npz")
    w = _sub(z1, "w/")
    x = torch.from_numpy(z1["x"])
    y = torch.from_numpy(z1["y"].astype(np.int64))
    keeps = _sub(z, "mask/")
    assert len(keeps) == 8
    loss, logits, grads = O.train_step(w, x, y, 2, keeps=keeps)
    assert (logits - torch.from_numpy(z["logits"])).abs().max().item() <= 2e-5
    assert abs(loss.item() - float(z["loss"])) <= 1e-6
    for k, b in _sub(z, "g/").items():
        assert (grads[k] - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), k


OPS = {
    "res_encode_c8": lambda x, w: O.res_block(x[0], w, ""),
    "res_pool_c8_16": lambda x, w: O.res_block(x[0], w, "", stride=2),
    "res_pool_odd_c3_8": lambda x, w: O.res_block(x[0], w, "", stride=2),
    "res_decode_c16_8": lambda x, w: O.res_block(x[0], w, ""),
    "res_encode_c30": lambda x, w: O.res_block(x[0], w, ""),
    "res_encode_c32": lambda x, w: O.res_block(x[0], w, ""),
    "stack3_c8": lambda x, w: O.res_stack(x[0], w, "", 3),
    "convtrans_c16_8": lambda x, w: O.conv_trans(x[0], w, ""),
    "convtrans_c32_16": lambda x, w: O.conv_trans(x[0], w, ""),
    "upconcat_c16_8": lambda x, w: O.up_concat(x[0], x[1], w, ""),
    "stem_c1_8": lambda x, w: torch.nn.functional.conv3d(x[0], w["weight"], w["bias"], padding=1),
    "stem_c3_32": lambda x, w: torch.nn.functional.conv3d(x[0], w["weight"], w["bias"], padding=1),
    "head_c8_3": lambda x, w: torch.nn.functional.conv3d(x[0], w["weight"], w["bias"]),
    "head_c32_4": lambda x, w: torch.nn.functional.conv3d(x[0], w["weight"], w["bias"]),
    "skip_k1s2_c8_16": lambda x, w: torch.nn.functional.conv3d(x[0], w["weight"], w["bias"], stride=2),
}


@pytest.mark.parametrize("tag", sorted(OPS))
def test_g3_ops(golden_dir, tag):
    z = _load(golden_dir, "g3_ops.npz")
    w = {k: v.requires_grad_(True) for k, v in _sub(z, tag + "/w/").items()}
    xs = []
    i = 0
    while "%s/in%d" % (tag, i) in z.files:
        xs.append(torch.from_numpy(z["%s/in%d" % (tag, i)]).requires_grad_(True))
        i += 1
    out = OPS[tag](xs, w)
    ref = torch.from_numpy(z[tag + "/out"])
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-5
    out.backward(torch.from_numpy(z[tag + "/gout"]))
    for i, xi in enumerate(xs):
        r = torch.from_numpy(z["%s/gin%d" % (tag, i)])
        assert (xi.grad - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())
    for k, r in _sub(z, tag + "/g/").items():
        assert (w[k].grad - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item()), k


LOSSES = {
    "hybird_iia": lambda x, y: O.hybird_loss(x, y, weight_v=[1.1, 11.6, 205.8, 466.8], alpha=0.9, beta=0.1),
    "hybird_default": lambda x, y: O.hybird_loss(x, y),
    "hybird_gamma3": lambda x, y: O.hybird_loss(x, y, gamma=3, weight_v=[1, 10, 20, 5]),
    "diceloss_iia": lambda x, y: O.dice_loss(x, y, weight_v=[1.1, 11.6, 205.8, 466.8], alpha=0.9, beta=0.1),
    "diceloss_default": lambda x, y: O.dice_loss(x, y),
    "focal_iia": lambda x, y: O.focal_loss(x, y, weight_v=[1.1, 11.6, 205.8, 466.8]),
    "focal_default": lambda x, y: O.focal_loss(x, y),
    "dice_kd": lambda x, y: O.dice_metric(x, y, weight_v=[0, 1, 0, 0]),
    "dice_default": lambda x, y: O.dice_metric(x, y),
    "dice_tversky": lambda x, y: O.dice_metric(x, y, weight_v=[1.1, 11.6, 205.8, 466.8], alpha=0.3, beta=0.7),
}


@pytest.mark.parametrize("tag", sorted(LOSSES))
def test_g3_loss(golden_dir, tag):
    z = _load(golden_dir, "g3_loss.npz")
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = torch.from_numpy(z["y"].astype(np.int64))
    v = LOSSES[tag](x, y)
    ref = float(z[tag + "/value"])
    assert abs(v.item() - ref) <= 3e-7 * max(1.0, abs(ref))
    v.backward()
    g = torch.from_numpy(z[tag + "/grad"])
    assert (x.grad - g).abs().max().item() <= 1e-6 * max(1.0, g.abs().max().item()) + 1e-9


def test_g3_loss_kits_and_functional_dice(golden_dir):
    z = _load(golden_dir, "g3_loss.npz")
    x = torch.from_numpy(z["kits/x"]).requires_grad_(True)
    y = torch.from_numpy(z["kits/y"].astype(np.int64))
    v = O.hybird_loss(x, y, weight_v=[1, 10, 20])
    assert abs(v.item() - float(z["kits/value"])) <= 3e-7
    p = torch.from_numpy(z["fdice/p"])
    g = torch.from_numpy(z["fdice/g"].astype(np.int64))
    assert abs(O.tversky(p, g).item() - float(z["fdice/default"])) <= 1e-7
    assert abs(O.tversky(p, g, 0.9, 0.1).item() - float(z["fdice/a9b1"])) <= 1e-7


def test_g4_quirks(golden_dir):
    q = json.load(open(os.path.join(golden_dir, "g4_quirks.json")))
    assert q["weight_c_ignored_hybird"][2] and q["weight_c_ignored_focal"][2] and q["weight_c_ignored_diceloss"][2]
    x = O.synth_image((2, 3, 6, 6, 6), 77)
    y = O.phantom_labels(2, (6, 6, 6), 3)
    assert abs(O.hybird_loss(x, y, weight_v=[1, 10, 20]).item() - q["weight_c_ignored_hybird"][0]) <= 3e-7
    y01 = y.clamp(max=1)
    assert abs(O.dice_loss(x, y01, weight_v=[0, 0, 1]).item() - q["absent_class_diceloss_w001"]) <= 1e-7
    assert abs(O.dice_metric(x, y01, weight_v=[0, 0, 1]).item() - q["absent_class_dice_w001"]) <= 1e-9
    assert q["c1_raises"]
    x1 = O.synth_image((1, 1, 4, 4, 4), 78)
    y1 = O.phantom_labels(1, (4, 4, 4), 2)
    with pytest.raises(RuntimeError):
        O.hybird_loss(x1, y1)
    assert abs(O.hybird_loss(x1, torch.zeros_like(y1)).item() - q["c1_all_zero_target_value"]) <= 3e-7


def test_g5_state_dict_contract(golden_dir):
    info = json.load(open(os.path.join(golden_dir, "g5_checkpoint.json")))
    w = O.init_state_dict(2, 8, 1, 2)
    assert {k: list(v.shape) for k, v in w.items()} == info["model_state_dict"]
    assert sum(v.numel() for v in w.values()) == info["num_parameters"]
    w2 = O.init_state_dict(4, 32, 1, 3)
    assert sum(v.numel() for v in w2.values()) == info["config2_num_parameters"] == 96783299
    assert O.paired_features(2, 8) == [[8, 8], [16, 16], [32, 32], [16, 16], [8, 8]]


def test_naive_primitives_pin_torch_conv():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 5, 6, 7, generator=g)
    w = torch.randn(4, 3, 3, 3, 3, generator=g)
    b = torch.randn(4, generator=g)
    for s in (1, 2):
        ref = torch.nn.functional.conv3d(x, w, b, stride=s, padding=1).numpy()
        assert np.abs(O.conv3d_naive(x, w, b, s, 1) - ref).max() < 1e-4
    w1 = torch.randn(4, 3, 1, 1, 1, generator=g)
    ref = torch.nn.functional.conv3d(x, w1, b, stride=2).numpy()
    assert np.abs(O.conv3d_naive(x, w1, b, 2, 0) - ref).max() < 1e-5
    wt = torch.randn(3, 5, 3, 3, 3, generator=g)
    bt = torch.randn(5, generator=g)
    ref = torch.nn.functional.conv_transpose3d(x, wt, bt, stride=2, padding=1).numpy()
    assert np.abs(O.conv_transpose3d_naive(x, wt, bt) - ref).max() < 1e-4


# --------------------------------------------------------------------------- G6: sliding-window inference
def _g6_case(g, tag):
    image = g[tag + "/image"]
    patch = tuple(int(v) for v in g[tag + "/patch"])
    spp, pool, feat, ncls = (int(v) for v in g[tag + "/meta"])
    w = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "/w/")}
    return image, patch, spp, pool, ncls, w


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_g6_predict_per_patch_matches_reference(golden_dir, tag):
    """trainer.predict_per_patch restated on the oracle forward: identical masks, identical NaN pattern of the
    never-visited border, probabilities within 2e-6 (fp32 softmax sums of <= 24 windows)."""
    g = np.load(os.path.join(golden_dir, "g6_predict.npz"))
    image, patch, spp, pool, ncls, w = _g6_case(g, tag)
    mask, _ = O.predict_per_patch(image, w, pool, ncls, patch, spp, False)
    prob, _ = O.predict_per_patch(image, w, pool, ncls, patch, spp, True)
    gm, gp = g[tag + "/mask"], g[tag + "/prob"]
    assert mask.dtype == np.uint8 and mask.shape == gm.shape == image.shape[:3]
    assert np.array_equal(mask, gm)
    assert prob.dtype == np.float32 and prob.shape == gp.shape
    assert np.array_equal(np.isnan(prob), np.isnan(gp))
    assert np.nanmax(np.abs(prob - gp)) < 2e-6
    if tag in "abd":
        assert np.isnan(gp).any()          # the quirk is really in the fixture: part of the far border is never visited
        assert (gm[np.isnan(gp).any(axis=-1)] == 0).all()


# --------------------------------------------------------------------------- G7: patch sampling / augmentation
G7_CASES = {
    "iia_like": dict(scale=0.1, crop_mode="random"),
    "iia_like_b": dict(scale=0.1, crop_mode="random"),
    "binary_label": dict(scale=0.2, crop_mode="random"),
    "pads": dict(scale=0.1, crop_mode="random"),
    "center_two_ch": dict(scale=[0.8, 1.3], crop_mode="center"),
    "margin_enforce": dict(scale=0.1, crop_mode="random", crop_margin=4, enforce_label_indices=[2]),
}


@pytest.mark.parametrize("tag", sorted(G7_CASES))
def test_g7_augment_oracle_matches_reference_pipeline(golden_dir, tag):
    """oracle/augment_oracle.pipeline under the fixture's numpy seed == the reference's transform pipeline
    (nb_train_iia.py:30-39) run by make_golden_augment.py: the same random draws in the same order, labels
    identical, resampled image within 2e-6 (float64 interpolation rounded to float32 on both sides), intensity chain
    within 1e-5 (float32 mean / power)."""
    from oracle import augment_oracle as A
    z = np.load(os.path.join(golden_dir, "g7_augment.npz"))
    np.random.seed(int(z[tag + "/seed"]))
    img, lab, after_crop, after_mirror = A.pipeline(z[tag + "/image_in"], z[tag + "/label_in"],
                                                    tuple(int(v) for v in z[tag + "/patch"]), **G7_CASES[tag])
    assert np.array_equal(lab, z[tag + "/label_out"])
    assert np.abs(after_crop - z[tag + "/after_crop"]).max() <= 2e-6
    assert np.abs(after_mirror - z[tag + "/after_mirror"]).max() <= 2e-6
    assert img.shape == z[tag + "/image_out"].shape and img.dtype == np.float32
    assert np.abs(img - z[tag + "/image_out"]).max() <= 1e-5
