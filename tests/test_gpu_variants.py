"""Attention-gated and BatchNorm model variants against fixtures produced by the reference's own
ResAttrUnet3D / ResAttrBNUnet3D (tests/golden/make_golden_variants.py).  The attention gate runs natively
(ops.AttGateFn: conv kernels + ru3d_pointwise); the BatchNorm variant runs its blocks as torch modules on the GPU
(BatchNorm is outside the native path) and is pinned here so that the fallback keeps the reference's numbers.
Run with `-m gpu`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import loss as L  # noqa: E402
import network  # noqa: E402
import _ops as ops  # noqa: E402

DEV = torch.device("cuda:0")


def _load(golden_dir, tag, ctor, train):
    z = np.load(os.path.join(golden_dir, "g8_variants.npz"))
    torch.manual_seed(0)
    model = ctor(num_pool=2, num_features=8, in_channels=1, out_channels=3)
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/w/")}
    model.load_state_dict(sd, strict=True)          # same state_dict keys as the reference
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout3d):
            m.p = 0.0
    model = model.to(DEV).train(train)
    x = torch.from_numpy(z["x"]).to(DEV)
    y = torch.from_numpy(z["y"].astype(np.int64)).to(DEV)
    return z, model, x, y


def _check(z, tag, model, logits, loss, gtol):
    ref = torch.from_numpy(z[tag + "/logits"])
    assert (logits.detach().cpu() - ref).abs().max().item() <= 2e-4
    assert abs(float(loss.detach()) - float(z[tag + "/loss"])) <= 1e-5
    checked = 0
    for k, p in model.named_parameters():
        key = "%s/g/%s" % (tag, k)
        if key not in z.files:
            assert p.grad is None, k
            continue
        want = torch.from_numpy(z[key])
        if p.grad is None:      # conv bias in front of a norm layer: identically zero gradient
            assert k.endswith(("conv1.bias", "conv2.bias")) and float(want.abs().max()) < 1e-5, k
            continue
        if k.endswith(("conv1.bias", "conv2.bias")) and float(want.abs().max()) < 1e-6:
            continue            # bias in front of a norm layer: analytically zero, both sides hold rounding noise
        # relative L2: single elements of these fp32 gradients are noisy - torch-ROCm against torch-CPU on the SAME
        # modules (the BatchNorm case below) lands up to 4.6 % of a tensor's maximum away on a pooling conv
        err = ((p.grad.cpu() - want).norm() / want.norm().clamp_min(1e-12)).item()
        assert err <= gtol, (k, err)
        checked += 1
    assert checked >= 20


def test_res_attr_unet_native_attention_gate_vs_reference(golden_dir):
    z, model, x, y = _load(golden_dir, "attr", network.ResAttrUnet3D, False)
    assert model.net.up_blocks[0].att_gate._native and model.net._native_chain() is not None
    calls = []
    orig = ops.AttGateFn.apply
    ops.AttGateFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        logits = model(x)
    finally:
        ops.AttGateFn.apply = orig
    assert len(calls) == 2                                   # both decoder levels went through the native gate
    loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
    loss.backward()
    _check(z, "attr", model, logits, loss, 3e-2)
    gate = model.net.up_blocks[1].att_gate.conv
    assert gate.weight.grad is not None and gate.bias.grad is not None


def test_res_attr_bn_unet_torch_fallback_vs_reference(golden_dir):
    z, model, x, y = _load(golden_dir, "attrbn", network.ResAttrBNUnet3D, True)
    assert model.net._native_chain() is None                 # BatchNorm blocks are torch modules
    logits = model(x)
    loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
    loss.backward()
    _check(z, "attrbn", model, logits, loss, 3e-2)
    for k, v in model.state_dict().items():                  # running statistics after one training forward
        key = "attrbn/after/" + k
        if key in z.files:
            assert np.allclose(v.cpu().numpy(), z[key], rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("feat,dtype", [(32, torch.bfloat16), (30, torch.bfloat16), (30, torch.float16)])
def test_attention_gate_16_bit_and_padded(feat, dtype):
    """The gate on the MFMA kernels (F = 32) and on channel-padded activations (F = 30): logits close to the fp32 run
    of the same weights, finite gradients of the parameters' shapes."""
    torch.manual_seed(2)
    model = network.ResAttrUnet3D(2, feat, 1, 2).to(DEV).eval()
    x = torch.randn(1, 1, 32, 32, 32, device=DEV)
    y = (torch.rand(1, 32, 32, 32, device=DEV) > 0.6).long()
    with torch.no_grad():
        ref = model(x)
    network.set_compute_dtype(model, dtype)
    assert model.net._pad == (feat == 30)
    logits = model(x)
    assert (logits - ref).abs().max().item() <= (0.12 if dtype == torch.bfloat16 else 0.03)
    L.HybirdLoss()(logits, y).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert p.grad.shape == p.shape and torch.isfinite(p.grad).all(), k
    assert model.net.up_blocks[0].att_gate.conv.weight.grad.abs().max().item() > 0


def test_res_attr_bn_unet_inference_runs_native_and_matches_reference(golden_dir):
    """Inference mode (eval + no_grad) of the BatchNorm variant runs on the native kernels (BatchNorm with running
    statistics = per-channel affine through the norm-apply kernel, attention gate native): logits against the
    reference's eval-mode forward with the running statistics its training forward left behind."""
    z, model, x, y = _load(golden_dir, "attrbn", network.ResAttrBNUnet3D, False)
    sd = model.state_dict()
    for k in list(sd):
        key = "attrbn/after/" + k
        if key in z.files:
            sd[k] = torch.from_numpy(z[key]).to(sd[k].device)
    model.load_state_dict(sd)
    assert model.net._bn_blocks is not None and model.net._native_chain() is None
    calls = []
    orig = network.ResBlock._forward_bn_eval
    network.ResBlock._forward_bn_eval = lambda self, t: (calls.append(1), orig(self, t))[1]
    try:
        with torch.no_grad():
            logits = model(x)
    finally:
        network.ResBlock._forward_bn_eval = orig
    assert len(calls) == 8                                  # 4 encode + 2 pool + 2 decode blocks, all native
    ref = torch.from_numpy(z["attrbn/logits_eval"])
    assert (logits.cpu() - ref).abs().max().item() <= 2e-4
    assert (logits.argmax(1).cpu() != ref.argmax(1)).float().mean().item() < 1e-4
    # with the tape on, the same call goes through the torch modules and still matches
    logits_t = model(x)
    assert len(calls) == 8 and (logits_t.detach().cpu() - ref).abs().max().item() <= 2e-4


@pytest.mark.parametrize("feat,dtype", [(32, torch.bfloat16), (30, torch.bfloat16)])
def test_bn_inference_16_bit_and_padded(feat, dtype):
    torch.manual_seed(4)
    model = network.ResAttrBNUnet3D(2, feat, 1, 2).to(DEV)
    x = torch.randn(2, 1, 32, 32, 32, device=DEV)
    model.train()
    with torch.no_grad():
        model(x)                                            # one training forward: non-trivial running statistics
    model.eval()
    with torch.no_grad():
        ref = model(x)
        network.set_compute_dtype(model, dtype)
        got = model(x)
    assert model.net._pad == (feat == 30)
    assert (got - ref).abs().max().item() <= 0.15
    model.train()
    out = model(x)                                          # back to the torch modules: un-padded again
    assert not model.net._pad and torch.isfinite(out).all()
