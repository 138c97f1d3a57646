#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference itself*.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference's own `network.py` / `loss.py`
(torch-only imports) under torch CPU fp32 and freezes inputs + outputs into
small .npz files.  Nothing from the reference's source text is stored: the
fixtures are tensors (inputs, weights, expected outputs) only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Groups:
  g1_config1.npz   whole-net, config 1, eval mode: weights, input, labels, logits,
                   argmax, 4 loss values, all parameter grads, params after 1/3 Adam steps
  g2_dropout.npz   same net in train mode: recovered Dropout3d keep masks, logits, loss, grads
  g3_ops.npz       per-block micro cases (ResBlock x3 roles, ConvTrans3D, UpConcat, stem/head)
  g3_loss.npz      each loss/metric with nb_train_iia.py hyper-parameters: value + d/dlogits
  g4_quirks.json   quirk pins (weight_c ignored, absent-class weighting, C==1 raises)
  g5_checkpoint.json  key names/shapes of a Trainer.save_checkpoint-shaped dict
  g6_predict.npz   trainer.predict_per_patch (sliding-window inference): volumes, weights, uint8 masks
                   and one-hot probability maps, incl. the uncovered-border (0/0 -> class 0) quirk

G6 imports the reference's trainer.py.  Its module header imports third-party packages that are not
installed here (apex, torchsummary, tensorboard, nibabel, transforms3d) and that predict_per_patch
never touches; those names are registered as empty placeholder modules for the import only.  The
function also uses `np.int`, an alias of the builtin `int` that numpy removed in 1.24; the alias is
restored for the duration of the call.  transform.py (numpy + scipy) is imported for real.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("RU3D_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_network = _load("network")
ref_loss = _load("loss")


def synth_image(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).clamp_(-2.34, 2.64)


def phantom_labels(n, dims, num_classes):
    """Background 0, ellipsoid of class 1, small spheres of class 2 (and 3) inside it."""
    d, h, w = dims
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    lab = np.zeros((n, d, h, w), dtype=np.int64)
    for i in range(n):
        cz, cy, cx = d * (0.5 + 0.05 * i), h * 0.5, w * (0.45 + 0.05 * i)
        e = ((zz - cz) / (0.30 * d)) ** 2 + ((yy - cy) / (0.25 * h)) ** 2 + ((xx - cx) / (0.22 * w)) ** 2
        lab[i][e <= 1.0] = 1
        if num_classes > 2:
            s = (zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2
            lab[i][s <= (0.09 * min(dims)) ** 2] = 2
        if num_classes > 3:
            s = (zz - cz - 0.15 * d) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2
            lab[i][s <= (0.06 * min(dims)) ** 2] = 3
    return torch.from_numpy(lab)


def sd_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def grads_np(model, prefix):
    out = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[prefix + k] = p.grad.detach().cpu().numpy().copy()
    return out


# --------------------------------------------------------------------------- G1
def make_g1():
    torch.manual_seed(0)
    model = ref_network.ResUnet3D(num_pool=2, num_features=8, in_channels=1, out_channels=2)
    x = synth_image((1, 1, 32, 32, 32), 1234)
    y = phantom_labels(1, (32, 32, 32), 2)
    out = {"x": x.numpy(), "y": y.numpy().astype(np.uint8)}
    out.update(sd_np(model.state_dict(), "w/"))

    model.eval()
    logits = model(x)
    out["logits"] = logits.detach().numpy().copy()
    out["argmax"] = logits.argmax(1).numpy().astype(np.uint8)
    losses = {
        "hybird": ref_loss.HybirdLoss(),
        "diceloss": ref_loss.DiceLoss(),
        "focal": ref_loss.FocalLoss(),
        "dice": ref_loss.Dice(),
    }
    for k, fn in losses.items():
        out["loss/" + k] = np.float32(fn(logits, y).item())
    loss = losses["hybird"](logits, y)
    loss.backward()
    out.update(grads_np(model, "g/"))
    none_grads = [k for k, p in model.named_parameters() if p.grad is None]
    out["none_grad_keys"] = np.array(none_grads)

    # Adam(lr=1e-4) steps in eval mode (dropout off), state after 1 and 3 steps
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    step_losses = []
    for step in range(3):
        opt.zero_grad()
        l = losses["hybird"](model(x), y)
        l.backward()
        opt.step()
        step_losses.append(l.item())
        if step == 0:
            out.update(sd_np(model.state_dict(), "adam1/"))
    out.update(sd_np(model.state_dict(), "adam3/"))
    out["adam_losses"] = np.array(step_losses, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "g1_config1.npz"), **out)
    print("g1: %d arrays, hybird=%.7f, none-grad params=%d" % (len(out), out["loss/hybird"], len(none_grads)))


# --------------------------------------------------------------------------- G2
def make_g2():
    torch.manual_seed(0)
    model = ref_network.ResUnet3D(num_pool=2, num_features=8, in_channels=1, out_channels=2)
    x = synth_image((1, 1, 32, 32, 32), 1234)
    y = phantom_labels(1, (32, 32, 32), 2)
    masks = {}

    def hook_for(name):
        def hook(mod, inp, outp):
            # a channel is dropped <=> whole (n,c) volume is exactly zero
            keep = (outp.detach().abs().amax(dim=(2, 3, 4)) > 0).to(torch.uint8)
            masks[name] = keep.numpy()
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Dropout3d):
            mod.register_forward_hook(hook_for(name))
    model.train()
    torch.manual_seed(7)
    logits = model(x)
    loss = ref_loss.HybirdLoss()(logits, y)
    loss.backward()
    out = {"logits": logits.detach().numpy(), "loss": np.float32(loss.item())}
    for k, v in masks.items():
        out["mask/" + k] = v
    out.update(grads_np(model, "g/"))
    np.savez_compressed(os.path.join(OUT, "g2_dropout.npz"), **out)
    print("g2: %d dropout sites, loss=%.7f" % (len(masks), out["loss"]))


# --------------------------------------------------------------------------- G3 ops
def run_block(tag, mod, inputs, out, train=False):
    """Forward + backward of one reference block with a fixed random upstream grad."""
    mod.train(train)
    ins = [t.clone().requires_grad_(True) for t in inputs]
    yv = mod(*ins)
    g = torch.Generator().manual_seed(99)
    gy = torch.randn(yv.shape, generator=g)
    yv.backward(gy)
    for i, t in enumerate(inputs):
        out["%s/in%d" % (tag, i)] = t.numpy()
        out["%s/gin%d" % (tag, i)] = ins[i].grad.numpy()
    out["%s/out" % tag] = yv.detach().numpy()
    out["%s/gout" % tag] = gy.numpy()
    out.update(sd_np(mod.state_dict(), tag + "/w/"))
    out.update(grads_np(mod, tag + "/g/"))


def make_g3_ops():
    out = {}
    torch.manual_seed(3)
    # dropout_op=None => deterministic train-mode forward (InstanceNorm has no state)
    cases = [
        ("res_encode_c8", lambda: ref_network.ResBlock(8, 8), [(2, 8, 6, 8, 10)]),
        ("res_pool_c8_16", lambda: ref_network.ResBlock(8, 16, stride=2), [(2, 8, 8, 8, 12)]),
        ("res_pool_odd_c3_8", lambda: ref_network.ResBlock(3, 8, stride=2), [(1, 3, 7, 9, 11)]),
        ("res_decode_c16_8", lambda: ref_network.ResBlock(16, 8), [(2, 16, 6, 6, 8)]),
        ("res_encode_c30", lambda: ref_network.ResBlock(30, 30), [(1, 30, 5, 6, 7)]),
        ("res_encode_c32", lambda: ref_network.ResBlock(32, 32), [(1, 32, 8, 8, 8)]),
        ("stack3_c8", lambda: ref_network.ResBlockStack(8, 8, num_stacks=3), [(1, 8, 6, 6, 6)]),
        ("convtrans_c16_8", lambda: ref_network.ConvTrans3D(16, 8), [(2, 16, 4, 5, 6)]),
        ("convtrans_c32_16", lambda: ref_network.ConvTrans3D(32, 16), [(1, 32, 4, 4, 4)]),
        ("upconcat_c16_8", lambda: ref_network.UpConcat(16, 8), [(2, 16, 3, 4, 5), (2, 8, 6, 8, 10)]),
        ("stem_c1_8", lambda: torch.nn.Conv3d(1, 8, kernel_size=3, padding=1), [(2, 1, 8, 9, 10)]),
        ("stem_c3_32", lambda: torch.nn.Conv3d(3, 32, kernel_size=3, padding=1), [(1, 3, 6, 6, 6)]),
        ("head_c8_3", lambda: torch.nn.Conv3d(8, 3, kernel_size=1), [(2, 8, 6, 7, 8)]),
        ("head_c32_4", lambda: torch.nn.Conv3d(32, 4, kernel_size=1), [(1, 32, 4, 4, 4)]),
        ("skip_k1s2_c8_16", lambda: torch.nn.Conv3d(8, 16, kernel_size=1, stride=2), [(1, 8, 7, 8, 9)]),
    ]
    for i, (tag, ctor, shapes) in enumerate(cases):
        mod = ctor()
        inputs = [synth_image(s, 100 + 10 * i + j) for j, s in enumerate(shapes)]
        run_block(tag, mod, inputs, out, train=False)
    np.savez_compressed(os.path.join(OUT, "g3_ops.npz"), **out)
    print("g3_ops: %d arrays" % len(out))


# --------------------------------------------------------------------------- G3 loss
def make_g3_loss():
    out = {}
    wc = [1, 1, 2, 2.9]
    wv = [1.1, 11.6, 205.8, 466.8]
    x = synth_image((2, 4, 10, 12, 14), 555) * 2.0
    y = phantom_labels(2, (10, 12, 14), 4)
    out["x"] = x.numpy()
    out["y"] = y.numpy().astype(np.uint8)
    cases = {
        "hybird_iia": ref_loss.HybirdLoss(weight_c=wc, weight_v=wv, alpha=0.9, beta=0.1),
        "hybird_default": ref_loss.HybirdLoss(),
        "hybird_gamma3": ref_loss.HybirdLoss(gamma=3, weight_v=[1, 10, 20, 5]),
        "diceloss_iia": ref_loss.DiceLoss(weight_c=wc, weight_v=wv, alpha=0.9, beta=0.1),
        "diceloss_default": ref_loss.DiceLoss(),
        "focal_iia": ref_loss.FocalLoss(weight_c=wc, weight_v=wv),
        "focal_default": ref_loss.FocalLoss(),
        "dice_kd": ref_loss.Dice(weight_v=[0, 1, 0, 0]),
        "dice_default": ref_loss.Dice(),
        "dice_tversky": ref_loss.Dice(weight_v=wv, alpha=0.3, beta=0.7),
    }
    for k, fn in cases.items():
        xi = x.clone().requires_grad_(True)
        v = fn(xi, y)
        v.backward()
        out[k + "/value"] = np.float32(v.item())
        out[k + "/grad"] = xi.grad.numpy()
    # 3-class KiTS19-style case on a different shape (nb_train_KITS19.py:20 weights)
    x3 = synth_image((1, 3, 9, 9, 9), 556) * 3.0
    y3 = phantom_labels(1, (9, 9, 9), 3)
    xi = x3.clone().requires_grad_(True)
    v = ref_loss.HybirdLoss(weight_v=[1, 10, 20])(xi, y3)
    v.backward()
    out["kits/x"] = x3.numpy()
    out["kits/y"] = y3.numpy().astype(np.uint8)
    out["kits/value"] = np.float32(v.item())
    out["kits/grad"] = xi.grad.numpy()
    # the functional `dice` used by trainer.evaluate_case on hard masks
    p = (torch.rand(3, 8, 8, 8, generator=torch.Generator().manual_seed(5)) > 0.5).float()
    g = (torch.rand(3, 8, 8, 8, generator=torch.Generator().manual_seed(6)) > 0.5).long()
    out["fdice/p"] = p.numpy()
    out["fdice/g"] = g.numpy().astype(np.uint8)
    out["fdice/default"] = np.float32(ref_loss.dice(p, g).item())
    out["fdice/a9b1"] = np.float32(ref_loss.dice(p, g, alpha=0.9, beta=0.1).item())
    np.savez_compressed(os.path.join(OUT, "g3_loss.npz"), **out)
    print("g3_loss: %d arrays" % len(out))


# --------------------------------------------------------------------------- G4
def make_g4():
    q = {}
    x = synth_image((2, 3, 6, 6, 6), 77)
    y = phantom_labels(2, (6, 6, 6), 3)
    a = ref_loss.HybirdLoss(weight_c=[1, 1, 1], weight_v=[1, 10, 20])(x, y).item()
    b = ref_loss.HybirdLoss(weight_c=[5, 0.1, 7], weight_v=[1, 10, 20])(x, y).item()
    q["weight_c_ignored_hybird"] = [a, b, a == b]
    a = ref_loss.FocalLoss(weight_c=[1, 1, 1], weight_v=[1, 10, 20])(x, y).item()
    b = ref_loss.FocalLoss(weight_c=[5, 0.1, 7], weight_v=[1, 10, 20])(x, y).item()
    q["weight_c_ignored_focal"] = [a, b, a == b]
    a = ref_loss.DiceLoss(weight_c=[1, 1, 1], weight_v=[1, 10, 20])(x, y).item()
    b = ref_loss.DiceLoss(weight_c=[5, 0.1, 7], weight_v=[1, 10, 20])(x, y).item()
    q["weight_c_ignored_diceloss"] = [a, b, a == b]
    # absent class: target has no class 2 -> dice_2 = s/(0+beta*fp+s) ~ 0 -> DiceLoss ~ 1.0
    y01 = y.clamp(max=1)
    q["absent_class_diceloss_w001"] = ref_loss.DiceLoss(weight_v=[0, 0, 1])(x, y01).item()
    q["absent_class_dice_w001"] = ref_loss.Dice(weight_v=[0, 0, 1])(x, y01).item()
    # C == 1: F.one_hot(target, 1) raises for labels {0,1}
    x1 = synth_image((1, 1, 4, 4, 4), 78)
    y1 = phantom_labels(1, (4, 4, 4), 2)
    try:
        ref_loss.HybirdLoss()(x1, y1)
        q["c1_raises"] = False
    except Exception as e:  # noqa
        q["c1_raises"] = True
        q["c1_exception_type"] = type(e).__name__
    # all-background target with C==1 does not raise
    try:
        v = ref_loss.HybirdLoss()(x1, torch.zeros_like(y1)).item()
        q["c1_all_zero_target_value"] = v
    except Exception as e:  # noqa
        q["c1_all_zero_target_value"] = "raises:" + type(e).__name__
    q["x_seed"] = 77
    with open(os.path.join(OUT, "g4_quirks.json"), "w") as f:
        json.dump(q, f, indent=1)
    print("g4:", q)


# --------------------------------------------------------------------------- G5
def make_g5():
    torch.manual_seed(0)
    model = ref_network.ResUnet3D(num_pool=2, num_features=8, in_channels=1, out_channels=2)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.2, patience=25)
    info = {
        # key names of the dict written by the reference trainer (trainer.py:606-619)
        "checkpoint_keys": ["model_state_dict", "optimizer_state_dict", "current_epoch",
                            "train_indices", "valid_indices", "best_result"],
        "optional_keys": ["scheduler_state_dict", "amp_state_dict"],
        "model_state_dict": {k: list(v.shape) for k, v in model.state_dict().items()},
        "optimizer_param_groups": [{k: v for k, v in g.items() if k != "params"} | {"num_params": len(g["params"])}
                                   for g in opt.state_dict()["param_groups"]],
        "scheduler_state_keys": sorted(sch.state_dict().keys()),
        "num_parameters": int(sum(p.numel() for p in model.parameters())),
        "config2_num_parameters": int(sum(p.numel() for p in ref_network.ResUnet3D(4, 32, 1, 3).parameters())),
        "default_ctor": {"num_pool": 4, "num_features": 30, "in_channels": 1, "out_channels": 1},
    }
    with open(os.path.join(OUT, "g5_checkpoint.json"), "w") as f:
        json.dump(info, f, indent=1, default=str)
    print("g5: %d state_dict keys, %d params" % (len(info["model_state_dict"]), info["num_parameters"]))


# --------------------------------------------------------------------------- G6
def _load_ref_trainer():
    import types
    placeholders = {
        "apex": {"amp": None}, "torchsummary": {"summary": None}, "nibabel": {},
        "torch.utils.tensorboard": {"SummaryWriter": None},
        "data": {k: None for k in ("CaseDataset", "load_case", "save_pred", "orient_crop_case",
                                   "regions_crop_case", "resample_normalize_case")},
    }
    saved = {}
    for name, attrs in placeholders.items():
        saved[name] = sys.modules.get(name)
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.modules["loss"] = ref_loss
    sys.modules["transform"] = _load("transform")
    try:
        return _load("trainer")
    finally:
        for name, old in saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
        sys.modules.pop("loss", None)
        sys.modules.pop("transform", None)


def make_g6():
    ref_trainer = _load_ref_trainer()
    had = hasattr(np, "int")
    if not had:
        np.int = int
    out = {}
    cases = [
        # tag, volume shape, patch, step_per_patch, (num_pool, F, out_channels)
        ("a", (21, 19, 13), (16, 16, 8), 4, (2, 4, 3)),      # ragged steps, uncovered far border
        ("b", (11, 24, 10), (16, 16, 8), 2, (2, 4, 3)),      # axis 0 shorter than the patch by an odd amount:
                                                             # padded 3 in front, cropped back at offset 2
        ("c", (16, 16, 8), (16, 16, 8), 4, (2, 4, 3)),       # exactly one patch (step_size 0 -> 9999999)
        ("d", (20, 17, 9), (16, 16, 8), 4, (1, 4, 1)),       # single class: sigmoid, no second softmax
    ]
    try:
        for tag, shape, patch, spp, (pool, feat, ncls) in cases:
            torch.manual_seed(60 + ord(tag))
            model = ref_network.ResUnet3D(num_pool=pool, num_features=feat, in_channels=1, out_channels=ncls)
            vol = synth_image(shape + (1,), 600 + ord(tag)).numpy().copy()
            out[tag + "/image"] = vol
            out[tag + "/patch"] = np.array(patch)
            out[tag + "/meta"] = np.array([spp, pool, feat, ncls])
            out.update(sd_np(model.state_dict(), tag + "/w/"))
            mask = ref_trainer.predict_per_patch(vol.copy(), model, ncls, patch, spp, False, False)
            prob = ref_trainer.predict_per_patch(vol.copy(), model, ncls, patch, spp, False, True)
            out[tag + "/mask"] = np.asarray(mask).copy()
            out[tag + "/prob"] = np.asarray(prob).copy()
            print("g6/%s: mask %s %s classes %s, prob %s nan=%d" % (
                tag, mask.shape, mask.dtype, np.unique(mask).tolist(), prob.shape, int(np.isnan(prob).sum())))
    finally:
        if not had:
            del np.int
    np.savez_compressed(os.path.join(OUT, "g6_predict.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_g1()
    make_g2()
    make_g3_ops()
    make_g3_loss()
    make_g4()
    make_g5()
    make_g6()
