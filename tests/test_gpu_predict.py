"""Sliding-window inference (`trainer.predict_per_patch`) on the HIP path against the reference's own outputs
(tests/golden/g6_predict.npz, made by the reference's function) and against the CPU oracle.  `-m gpu` only.

Tolerances: averaged probabilities within 5e-6 abs in fp32 mode (logits agree to ~1e-5, a softmax output moves
by at most a quarter of the logit error, and up to 24 windows are averaged); the never-visited border is NaN in
the same voxels; masks are identical except where the reference's own top-2 probability margin is below 1e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import _native as N  # noqa: E402
import inference as I  # noqa: E402
import network  # noqa: E402
import trainer as T  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


def _case(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "g6_predict.npz"))
    patch = tuple(int(v) for v in g[tag + "/patch"])
    spp, pool, feat, ncls = (int(v) for v in g[tag + "/meta"])
    model = network.ResUnet3D(num_pool=pool, num_features=feat, in_channels=1, out_channels=ncls)
    model.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "/w/")},
                          strict=True)
    return g, g[tag + "/image"], patch, spp, ncls, model.to(DEV)


def _margin(prob):
    if prob.shape[-1] == 1:
        return np.abs(prob[..., 0] - 0.5)
    s = np.sort(np.nan_to_num(prob, nan=0.0), axis=-1)
    return s[..., -1] - s[..., -2]


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
@pytest.mark.parametrize("patch_batch", [1, 3])
def test_predict_per_patch_vs_reference_fixture(golden_dir, tag, patch_batch):
    g, image, patch, spp, ncls, model = _case(golden_dir, tag)
    gm, gp = g[tag + "/mask"], g[tag + "/prob"]
    prob = T.predict_per_patch(image, model, ncls, patch, spp, False, True, patch_batch=patch_batch)
    mask = T.predict_per_patch(image, model, ncls, patch, spp, False, False, patch_batch=patch_batch)
    assert prob.dtype == np.float32 and prob.shape == gp.shape
    assert np.array_equal(np.isnan(prob), np.isnan(gp))
    assert np.nanmax(np.abs(prob - gp)) < 5e-6
    assert mask.dtype == np.uint8 and mask.shape == gm.shape
    diff = mask != gm
    assert not (diff & (_margin(gp) > 1e-5)).any()
    assert diff.mean() < 1e-3
    assert (mask[np.isnan(gp).any(axis=-1)] == 0).all()


def test_patch_batch_does_not_change_the_result(golden_dir):
    g, image, patch, spp, ncls, model = _case(golden_dir, "a")
    a = T.predict_per_patch(image, model, ncls, patch, spp, False, True, patch_batch=1)
    b = T.predict_per_patch(image, model, ncls, patch, spp, False, True, patch_batch=5)
    assert np.array_equal(a, b, equal_nan=True)


def test_predict_kernels_against_numpy_on_random_windows():
    """ru3d_predict_accumulate / ru3d_predict_merge alone: random logits, overlapping windows, 1..4 classes,
    fp32 and bf16 logits, a crop offset; compared with the same arithmetic in numpy/torch on the CPU."""
    import ctypes
    rng = np.random.default_rng(3)
    X, Y, Z = 13, 11, 17
    for C in (1, 2, 3, 4):
        for dt in (torch.float32, torch.bfloat16):
            acc = torch.zeros((X, Y, Z, C), device=DEV)
            cnt = torch.zeros((X, Y, Z), device=DEV)
            racc = torch.zeros((X, Y, Z, C))
            rcnt = torch.zeros((X, Y, Z))
            for (ox, oy, oz) in ((0, 0, 0), (3, 1, 5), (5, 3, 9), (3, 1, 5)):
                z = torch.from_numpy(rng.standard_normal((2, 8, 8, 8, C)).astype(np.float32) * 3).to(dt)
                zd = z.to(DEV).permute(0, 4, 1, 2, 3)
                d = N.desc(zd)
                N.check(N.lib.ru3d_predict_accumulate(ctypes.byref(d), N.dtype_code(dt), 1, N.ptr(acc), N.ptr(cnt),
                                                      X, Y, Z, ox, oy, oz, N.stream()))
                zz = z[1].float()
                p = torch.sigmoid(zz) if C == 1 else torch.softmax(zz, dim=-1)
                racc[ox:ox + 8, oy:oy + 8, oz:oz + 8] += p
                rcnt[ox:ox + 8, oy:oy + 8, oz:oz + 8] += 1
            assert torch.equal(cnt.cpu(), rcnt)
            assert (acc.cpu() - racc).abs().max().item() < 1e-6
            crop, size = (1, 0, 2), (11, 10, 14)
            sl = tuple(slice(c, c + s) for c, s in zip(crop, size))
            prob = torch.empty(size + (C,), device=DEV)
            mask = torch.empty(size, dtype=torch.uint8, device=DEV)
            for one_hot, out in ((1, prob), (0, mask)):
                N.check(N.lib.ru3d_predict_merge(N.ptr(acc), N.ptr(cnt), X, Y, Z, C, *crop, *size, one_hot,
                                                 N.ptr(out), N.stream()))
            rp = (acc.cpu() / cnt.cpu()[..., None])[sl]
            assert np.array_equal(prob.cpu().numpy(), rp.numpy(), equal_nan=True)
            if C == 1:
                with np.errstate(invalid="ignore"):
                    rm = np.nan_to_num(np.round(rp[..., 0].numpy()), nan=0.0).astype(np.uint8)
            else:
                rm = torch.argmax(torch.softmax(rp, dim=-1), dim=-1).numpy().astype(np.uint8)
            got = mask.cpu().numpy()
            bad = got != rm
            assert not (bad & (_margin(rp.numpy()) > 1e-6)).any()
    # windows outside the volume and too many classes are refused, not launched
    z = torch.zeros((1, 8, 8, 8, 2), device=DEV).permute(0, 4, 1, 2, 3)
    d = N.desc(z)
    acc = torch.zeros((8, 8, 8, 2), device=DEV)
    cnt = torch.zeros((8, 8, 8), device=DEV)
    assert N.lib.ru3d_predict_accumulate(ctypes.byref(d), N.F32, 0, N.ptr(acc), N.ptr(cnt), 8, 8, 8, 1, 0, 0,
                                         N.stream()) != 0
    assert N.lib.ru3d_predict_accumulate(ctypes.byref(d), N.F32, 1, N.ptr(acc), N.ptr(cnt), 8, 8, 8, 0, 0, 0,
                                         N.stream()) != 0


def test_predict_config2_patch_bf16_vs_fp32_masks():
    """ResUnet3D(4,32,1,3), one 160x128x128 case, 128^3 windows, 2 steps per patch: the bf16 run's mask agrees
    with the fp32 run's on > 99.9 % of the confident voxels (random-init weights put most voxels near a tie, trained weights
    do not, so only voxels whose fp32 top-2 margin exceeds 0.05 are compared), both cover the same voxels, and the probabilities sum to 1 wherever a window reached."""
    torch.manual_seed(0)
    model = network.ResUnet3D(4, 32, 1, 3).to(DEV)
    image = O.synth_image((160, 128, 128, 1), 99).numpy()
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        network.set_compute_dtype(model, dt)
        out[dt] = T.predict_per_patch(image, model, 3, (128, 128, 128), 2, False, True, patch_batch=2)
    p32, p16 = out[torch.float32], out[torch.bfloat16]
    assert p32.shape == (160, 128, 128, 3)
    assert np.array_equal(np.isnan(p32), np.isnan(p16))
    ok = ~np.isnan(p32).any(axis=-1)
    assert ok.mean() > 0.9
    assert np.abs(p32[ok].sum(-1) - 1).max() < 1e-5
    assert np.abs(p32[ok] - p16[ok]).max() < 0.1
    sure = ok & (_margin(p32) > 0.05)
    agree = (p32[sure].argmax(-1) == p16[sure].argmax(-1)).mean()
    print("bf16 vs fp32: max prob diff %.4f, confident voxels %.3f, agreement there %.5f"
          % (np.abs(p32[ok] - p16[ok]).max(), sure.mean(), agree))
    assert agree > 0.999, agree


def test_packed_weight_cache_follows_weight_updates():
    """Inference reuses the packed bf16 weights between windows; the cache must notice every way weights change on this
    path: the fused optimizer (raw-pointer update), torch in-place ops / load_state_dict, and a new model that happens to
    be allocated where a freed one lived."""
    import _ops as ops
    import loss as L
    import optim
    torch.manual_seed(3)
    x = O.synth_image((1, 1, 32, 32, 32), 5).to(DEV)
    y = O.phantom_labels(1, (32, 32, 32), 3).to(DEV)

    def fresh(model):
        ops._PACK_CACHE.clear()
        with torch.no_grad():
            return model(x).clone()

    model = network.ResUnet3D(2, 32, 1, 3).to(DEV)
    network.set_compute_dtype(model, torch.bfloat16)
    model.eval()
    with torch.no_grad():
        a = model(x).clone()
        b = model(x).clone()                       # served from the cache
    assert torch.equal(a, b) and len(ops._PACK_CACHE) > 0
    opt = optim.Adam(model.parameters(), lr=1e-2)
    L.HybirdLoss()(model(x), y).backward()
    opt.step()                                     # fused kernel: parameters change behind torch's version counters
    with torch.no_grad():
        c = model(x).clone()
    assert not torch.equal(a, c) and torch.equal(c, fresh(model))
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.5)                            # torch in-place op
        d = model(x).clone()
    assert torch.equal(d, fresh(model))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    del model, opt
    other = network.ResUnet3D(2, 32, 1, 3).to(DEV)   # may land on the freed model's addresses
    network.set_compute_dtype(other, torch.bfloat16)
    other.eval()
    with torch.no_grad():
        e = other(x).clone()
    assert torch.equal(e, fresh(other))
    other.load_state_dict(sd)
    with torch.no_grad():
        f = other(x).clone()
    assert torch.equal(f, d)


# --------------------------------------------------------------------------- whole-case inference (trainer.py:101-133)
def test_predict_case_device_pipeline_vs_host_composition():
    """trainer.predict_case (resample + normalise + sliding window + resize, all on the device) against the same chain
    assembled from the host-side pieces the reference uses: data.resample_normalize_case (scipy zoom),
    predict_per_patch, transform.resize (scipy zoom, label rule)."""
    import data
    import network
    import trainer
    import transform as T
    torch.manual_seed(3)
    model = network.ResUnet3D(2, 8, 1, 3).to(DEV).eval()
    rng = np.random.RandomState(5)
    g = np.meshgrid(*[np.linspace(-1, 1, s) for s in (44, 40, 20)], indexing="ij")
    img = (80 * np.sin(3 * g[0]) * np.cos(2 * g[1]) + 60 * g[2] + 100 + 5 * rng.randn(44, 40, 20)).astype(np.float32)
    case = {"case_id": "c", "affine": np.diag([1.5, 1.5, 3.0, 1.0]), "image": img[..., None]}
    stats = {"mean": 100.0, "std": 60.0, "pct_00_5": -50.0, "pct_99_5": 250.0}
    spacing, patch = (1.0, 1.2, 2.0), (32, 32, 16)
    for one_hot in (False, True):
        got = trainer.predict_case(dict(case), model, spacing, stats, num_classes=3, patch_size=patch,
                                   step_per_patch=2, verbose=False, one_hot=one_hot)["pred"]
        rs = data.resample_normalize_case(case, spacing, stats)
        ref = trainer.predict_per_patch(rs["image"].astype(np.float32), model, 3, patch, 2, False, one_hot)
        want = T.resize(ref, case["image"].shape[:-1], is_label=one_hot is False)
        assert got.shape == want.shape
        if one_hot:
            ok = np.isfinite(want)
            assert np.array_equal(ok, np.isfinite(got))
            assert np.abs(got[ok] - want[ok]).max() <= 2e-4
        else:
            assert got.dtype == np.uint8
            assert (got != want).mean() <= 2e-3          # interpolation ulps can move a voxel that sits on a tie


# --------------------------------------------------------------------------- cascade (fixture G9)
def _g9_models(golden_dir):
    z = np.load(os.path.join(golden_dir, "g9_cascade.npz"))
    coarse = network.ResUnet3D(num_pool=2, num_features=4, in_channels=1, out_channels=1)
    detail = network.ResUnet3D(num_pool=2, num_features=4, in_channels=1, out_channels=3)
    coarse.load_state_dict({k[len("coarse/w/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("coarse/w/")}, strict=True)
    detail.load_state_dict({k[len("detail/w/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("detail/w/")}, strict=True)
    stats = dict(zip(("mean", "std", "pct_00_5", "pct_99_5"), (float(v) for v in z["stats"])))
    kw = dict(coarse_target_spacing=tuple(z["params"][0]), coarse_normalize_stats=stats,
              coarse_patch_size=tuple(int(v) for v in z["patches"][0]), detail_target_spacing=tuple(z["params"][1]),
              detail_normalize_stats=stats, detail_patch_size=tuple(int(v) for v in z["patches"][1]),
              step_per_patch=int(z["scalars"][0]), region_threshold=int(z["scalars"][1]), crop_padding=int(z["scalars"][2]))
    return z, coarse.to(DEV).eval(), detail.to(DEV).eval(), kw


def test_cascade_predict_case_vs_reference_fixture(golden_dir):
    """trainer.cascade_predict_case (reference trainer.py:164-245) on the HIP path against G9 - the reference's own
    coarse mask, regions and per-region probability maps, merged by the reference's arithmetic: the coarse mask agrees
    except on voxels within 1e-4 of the 0.5 threshold (none here), every region's probabilities within 2e-4 (the
    device resampling differs from scipy's by interpolation ulps, test_predict_case_device_pipeline_vs_host_composition),
    the final mask identical except where the merged top-2 margin is below 1e-3."""
    import data
    z, coarse, detail, kw = _g9_models(golden_dir)
    case = {"case_id": "g9", "image": z["image"], "affine": z["affine"]}
    c1 = T.predict_case(dict(case), coarse, kw["coarse_target_spacing"], kw["coarse_normalize_stats"], 1,
                        kw["coarse_patch_size"], kw["step_per_patch"], verbose=False)
    assert c1["pred"].dtype == np.uint8 and (c1["pred"] != z["coarse_pred"]).mean() <= 1e-3
    regions = data.regions_crop_case({**case, "pred": z["coarse_pred"]}, kw["region_threshold"], kw["crop_padding"], "pred")
    assert [r["bbox"].tolist() for r in regions] == z["regions"].tolist()
    for i, region in enumerate(regions):
        out = T.predict_case(region, detail, kw["detail_target_spacing"], kw["detail_normalize_stats"], 3,
                             kw["detail_patch_size"], kw["step_per_patch"], verbose=False, one_hot=True)["pred"]
        want = z["region%d/prob" % i]
        ok = np.isfinite(want)
        assert out.shape == want.shape and np.array_equal(ok, np.isfinite(out))
        assert np.abs(out[ok] - want[ok]).max() <= 2e-4
    res = T.cascade_predict_case(dict(case), coarse, kw["coarse_target_spacing"], kw["coarse_normalize_stats"],
                                 kw["coarse_patch_size"], detail, kw["detail_target_spacing"], kw["detail_normalize_stats"],
                                 kw["detail_patch_size"], 3, kw["step_per_patch"], kw["region_threshold"], kw["crop_padding"],
                                 verbose=False)
    assert res["pred"].dtype == np.uint8 and res["pred"].shape == z["pred"].shape
    assert sorted(np.unique(res["pred"]).tolist()) == sorted(np.unique(z["pred"]).tolist())
    assert (res["pred"] != z["pred"]).mean() <= 2e-3


def test_cascade_predict_from_files_restores_the_original_grid(golden_dir, tmp_path):
    """trainer.cascade_predict / batch_cascade_predict (reference trainer.py:248-345): the G9 volume stored with two
    flipped axes and an air margin comes back as a mask on the FILE's grid; inside the non-air box it is the cascade's
    mask of the reoriented crop, outside it is background."""
    import data
    import nifti
    z, coarse, detail, kw = _g9_models(golden_dir)
    vol = z["image"][..., 0]
    padded = np.full(tuple(s + 6 for s in vol.shape), -1000.0, dtype=np.float32)
    padded[2:-4, 3:-3, 1:-5] = vol
    flipped = padded[::-1, :, ::-1].copy()                       # stored right-to-left and top-to-bottom
    aff = np.diag([-1.6, 1.6, -3.0, 1.0])
    aff[:3, 3] = [40.0, 14.0, 60.0]
    idir = tmp_path / "images"
    idir.mkdir()
    nifti.save(flipped, aff, idir / "case_7.nii.gz")
    out = T.cascade_predict(idir / "case_7.nii.gz", coarse, kw["coarse_target_spacing"], kw["coarse_normalize_stats"],
                            kw["coarse_patch_size"], detail, kw["detail_target_spacing"], kw["detail_normalize_stats"],
                            kw["detail_patch_size"], air=-200, step_per_patch=kw["step_per_patch"],
                            region_threshold=kw["region_threshold"], crop_padding=kw["crop_padding"], verbose=False)
    assert out["pred"].shape == flipped.shape and out["pred"].dtype == np.uint8 and out["image"].shape == flipped.shape + (1,)
    # the same mask, computed by hand: reorient, crop, cascade, paste, orient back (a double flip is its own inverse)
    case = data.orient_crop_case({"case_id": "case_7", "image": flipped[..., None], "affine": aff}, -200)
    ref = T.cascade_predict_case(dict(case), coarse, kw["coarse_target_spacing"], kw["coarse_normalize_stats"],
                                 kw["coarse_patch_size"], detail, kw["detail_target_spacing"], kw["detail_normalize_stats"],
                                 kw["detail_patch_size"], 3, kw["step_per_patch"], kw["region_threshold"], kw["crop_padding"],
                                 verbose=False)["pred"]
    canon = np.zeros(padded.shape, dtype=np.uint8)
    bb = case["bbox"]
    canon[tuple(slice(b[0], b[1]) for b in bb)] = ref
    assert np.array_equal(out["pred"], canon[::-1, :, ::-1])
    assert out["pred"].max() >= 1 and not out["pred"][:, :, :4].any()     # the air margin stays background
    sdir = tmp_path / "preds"
    T.batch_cascade_predict(idir, sdir, coarse, kw["coarse_target_spacing"], kw["coarse_normalize_stats"],
                            kw["coarse_patch_size"], detail, kw["detail_target_spacing"], kw["detail_normalize_stats"],
                            kw["detail_patch_size"], air=-200, step_per_patch=kw["step_per_patch"],
                            region_threshold=kw["region_threshold"], crop_padding=kw["crop_padding"])
    saved, saved_aff, _ = nifti.load(sdir / "case_7.pred.nii.gz")
    assert np.array_equal(saved.astype(np.uint8), out["pred"]) and np.allclose(saved_aff, aff)
