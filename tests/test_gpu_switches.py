"""The kernel-selection switches (RU3D_*: INTEGRATION.md section 1) are read once per process, so the default suite only
ever runs the default combination (hygiene).  Here one forward + backward of the config-2 architecture
(reduced patch) and of the F = 30 / fp16 configuration on extents that fit no tile runs in fresh processes under the
non-default settings - every alternative kernel family behind a switch - and is held against the CPU oracle with the
default family's own distance as the yardstick; the BatchNorm variant's torch-module path is held against the native
one in fp32.  Run with `-m gpu`."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "switch_child.py")

SWITCHES = {
    "in": [{"RU3D_CONV_S2": "0", "RU3D_FUSED_SKIP": "0", "RU3D_DGRAD_PAIR": "0"},       # round-2 direct forms, unfused tails
           {"RU3D_CONV_SLIDE64": "0", "RU3D_WGRAD_SLIDE": "0", "RU3D_CONV_PC": "0"},    # no sliding 64-channel / wgrad kernels
           {"RU3D_CONV_WS": "2", "RU3D_STEM_MFMA": "0", "RU3D_HEAD_FUSED": "0"},         # whole-sample conv, VALU stem, unfused head backward
           {"RU3D_SKIP_LINK": "0", "RU3D_WGRAD_DIRECT": "0", "RU3D_WGRAD_STREAM": "auto", "RU3D_WGRAD_PAIR": "0"},  # concat copies, slabs at 8^3, weight gradients of the small levels on a second stream, skip-conv weight gradients as launches of their own
           # round 4: norm.hip's three launches on the small levels, the interleaved concat on the full-resolution level
           # ... and the deep levels on the 1 x 2-accumulator conv kernel instead of the split-in-workgroup one
           {"RU3D_IN_SMALL": "0", "RU3D_PLANAR_CONCAT": "0", "RU3D_CONV_SK": "0"}],
    "c4": [{"RU3D_CONV_TILEFIT": "0", "RU3D_SLIDE64_EDGE": "0", "RU3D_CONV_WS": "0"},    # round-2 tilings of the odd extents
           {"RU3D_PAD_CHANNELS": "0"}],                                                   # F = 30 on the generic kernels
}


def _run(tmp, tag, kind, env):
    out = os.path.join(tmp, tag + ".pt")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, CHILD, out, kind], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600)
    assert r.returncode == 0, r.stdout.decode("utf-8", "replace")[-3000:]
    return torch.load(out, weights_only=False)


def _oracle_grads(kind):
    """fp32 CPU oracle of the child's step (same seed -> same initial weights, same synthetic case, Dropout3d off)."""
    import network
    from oracle import unet_oracle as O
    torch.manual_seed(21)
    if kind == "c4":
        model, shape = network.ResUnet3D(4, 30, 1, 3), (2, 1, 80, 80, 48)
    else:
        model, shape = network.ResUnet3D(4, 32, 1, 3), (2, 1, 64, 64, 64)
    w0 = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
    x = O.synth_image(shape, 5)
    y = O.phantom_labels(shape[0], shape[2:], 3)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    loss, logits, grads = O.train_step(w0, x, y, 4, loss_kwargs={"weight_v": [1, 10, 20]})
    sim = None
    if kind == "in":        # the oracle's bf16 storage model: what 16-bit inter-kernel tensors cost by themselves
        O.set_storage(torch.bfloat16)
        try:
            _, _, gs = O.train_step(w0, x, y, 4, loss_kwargs={"weight_v": [1, 10, 20]})
        finally:
            O.set_storage(None)
        sim = {k: g.float() for k, g in gs.items()}
    return float(loss), logits.float(), {k: g.float() for k, g in grads.items()}, sim


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("kind", ["in", "c4"])
def test_non_default_kernel_switches_against_the_oracle(tmp_path, kind):
    """16-bit gradients of an untrained deep net sit far from the fp32 truth (0.3-0.7 relative L2 on the deep layers: the
    storage rounding is amplified through four levels of InstanceNorm + LeakyReLU - test_g1_whole_net_bf16 pins the
    same effect against the oracle's storage model), and two kernel families with different rounding points sit as far
    from each other.  So the yardstick for an alternative family is the default family's own distance to the oracle:
    no tensor may be more than 1.3x + 0.03 further away, and loss / logits must agree closely."""
    loss_o, logits_o, grads_o, grads_sim = _oracle_grads(kind)
    base = _run(str(tmp_path), "base", kind, {})
    assert abs(base["loss"] - loss_o) <= 5e-3
    e_base = {k: _rel(g, grads_o[k]) for k, g in base["grads"].items() if k.endswith("weight")}
    assert len(e_base) >= 50
    if grads_sim is not None:
        # the default family itself, on a shape where the stride-2 tile kernels, the pair kernels and the fused tails
        # engage (the fixtures' 32^3 nets are too small for them): no further from the truth than the storage model
        for k, eb in e_base.items():
            es = _rel(grads_sim[k], grads_o[k])
            assert eb <= 1.3 * es + 0.03, ("default kernels", k, eb, es)
    for i, env in enumerate(SWITCHES[kind]):
        alt = _run(str(tmp_path), "alt%d" % i, kind, env)
        what = "%s %s" % (kind, env)
        assert abs(alt["loss"] - loss_o) <= 5e-3, (what, alt["loss"], loss_o)
        assert (alt["logits"] - logits_o).abs().max().item() <= 1.5 * (base["logits"] - logits_o).abs().max().item() + 0.02, what
        assert set(alt["grads"]) == set(base["grads"]), what
        for k, eb in e_base.items():
            ea = _rel(alt["grads"][k], grads_o[k])
            assert ea <= 1.3 * eb + 0.03, (what, k, ea, eb)


def test_batchnorm_torch_modules_agree_with_native_training(tmp_path):
    """RU3D_BN_TRAIN=torch (BatchNorm blocks as torch modules) against the native path, training mode, fp32 storage on
    both sides and Dropout3d off (the two draw their masks from different generators): the same numbers."""
    base = _run(str(tmp_path), "base", "bn", {})
    alt = _run(str(tmp_path), "alt", "bn", {"RU3D_BN_TRAIN": "torch"})
    assert abs(alt["loss"] - base["loss"]) <= 2e-5
    assert (alt["logits"] - base["logits"]).abs().max().item() <= 5e-4
    assert set(alt["grads"]) == set(base["grads"])
    for k, g in alt["grads"].items():
        if k.endswith(("conv1.bias", "conv2.bias", "up.0.bias")) and g.abs().max().item() < 1e-5:
            continue                    # in front of a batch norm: analytically zero, rounding noise on both sides
        assert _rel(base["grads"][k], g) <= 3e-2, (k, _rel(base["grads"][k], g))
