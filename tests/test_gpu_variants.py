"""Attention-gated and BatchNorm model variants against fixtures produced by the reference's own
ResAttrUnet3D / ResAttrBNUnet3D (tests/golden/make_golden_variants.py).  The attention gate runs natively
(ops.AttGateFn: conv kernels + ru3d_pointwise); the BatchNorm variant trains on the native kernels too
(batch-pooled statistics: ops.ResBlockBNFn / UpBNFn) and is pinned here against the reference's own numbers.
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


def _count_calls(fn_cls):
    calls = []
    orig = fn_cls.apply
    fn_cls.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    return calls, (lambda: setattr(fn_cls, "apply", orig))


@pytest.mark.parametrize("mode", ["native", "torch"])
def test_res_attr_bn_unet_training_vs_reference(golden_dir, monkeypatch, mode):
    """One training forward + backward of the BatchNorm variant against the reference's own (fixture G8): logits, loss,
    every parameter gradient (gamma / beta included), running statistics and num_batches_tracked afterwards.
    native: the blocks run ops.ResBlockBNFn / UpBNFn (batch-pooled statistics on the HIP kernels); torch: the round-2
    path (RU3D_BN_TRAIN=torch, blocks as torch modules) as a cross-check of the fixture."""
    monkeypatch.setenv("RU3D_BN_TRAIN", mode)
    z, model, x, y = _load(golden_dir, "attrbn", network.ResAttrBNUnet3D, True)
    assert model.net._native_chain() is None and model.net._bn_blocks is not None
    res_calls, undo_res = _count_calls(ops.ResBlockBNFn)
    up_calls, undo_up = _count_calls(ops.UpBNFn)
    try:
        logits = model(x)
    finally:
        undo_res()
        undo_up()
    assert (len(res_calls), len(up_calls)) == ((8, 2) if mode == "native" else (0, 0))
    loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
    loss.backward()
    _check(z, "attrbn", model, logits, loss, 3e-2)
    seen = 0
    for k, v in model.state_dict().items():                  # running statistics after one training forward
        key = "attrbn/after/" + k
        if key in z.files:
            assert np.allclose(v.cpu().numpy(), z[key], rtol=1e-4, atol=1e-5), k
            seen += 1
    assert seen >= 30
    for k, p in model.named_parameters():
        if ".norm." in k or ".up.2." in k:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k


def _bn_block_oracle(block, x, keep, p):
    """fp64 CPU restatement of reference network.py:405-416 with a given Dropout3d keep mask."""
    import copy
    blk = copy.deepcopy(block).cpu().double()
    x = x.detach().cpu().double().requires_grad_(True)
    skip = blk.skip_conv(x) if blk.uses_skip_conv else x
    h = blk.conv1(x)
    if keep is not None:
        h = h * (keep.cpu().double() / (1.0 - p))[:, :, None, None, None]
    h = torch.nn.functional.leaky_relu(blk.norm(h), 0.01)
    h = blk.conv2(h)
    out = torch.nn.functional.leaky_relu(blk.norm(h) + skip, 0.01)
    return blk, x, out


@pytest.mark.parametrize("cin,cout,stride,drop", [(8, 8, 1, True), (8, 16, 2, True), (12, 8, 1, False)])
def test_bn_resblock_training_with_dropout_vs_fp64(cin, cout, stride, drop):
    """ResBlock(norm_op=BatchNorm3d) in training mode on the native kernels, fp32 storage, with a recorded Dropout3d
    mask: Dropout3d sits in FRONT of a batch norm, so the per-(n, c) factors enter the pooled statistics (a dropped
    sample's channel still counts as zeros) and conv1's bias keeps a gradient.  Against an fp64 restatement."""
    torch.manual_seed(5)
    blk = network.ResBlock(cin, cout, stride=stride, norm_op=torch.nn.BatchNorm3d).to(DEV).train()
    with torch.no_grad():
        blk.norm.weight.uniform_(0.5, 1.5)
        blk.norm.bias.uniform_(-0.3, 0.3)
    n = 3
    keep = None
    if drop:
        keep = (torch.rand(n, cout, device=DEV) > 0.4).float()
        keep[0, 0] = 0.0
        keep[1, 0] = 1.0
        blk._forced_keep = keep
    else:
        blk.dropout = None
    x = torch.randn(n, cin, 12, 10, 8, device=DEV, requires_grad=True)
    g = torch.randn(n, cout, 12 // stride, 10 // stride, 8 // stride, device=DEV)
    ref, xr, out_r = _bn_block_oracle(blk, x, keep, 0.5)
    out_r.backward(g.cpu().double())
    calls, undo = _count_calls(ops.ResBlockBNFn)
    try:
        out = blk(x)
    finally:
        undo()
    assert len(calls) == 1
    out.backward(g)
    assert (out.detach().cpu().double() - out_r.detach()).abs().max().item() <= 2e-4
    assert (x.grad.cpu().double() - xr.grad).abs().max().item() <= 2e-3 * max(1.0, xr.grad.abs().max().item())
    for (k, pn), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert pn.grad is None, k
            continue
        assert pn.grad is not None, k
        if k == "conv2.bias" or (k == "conv1.bias" and not drop):
            # straight in front of a batch norm: analytically zero, fp32 summation noise here
            assert pr.grad.abs().max().item() < 1e-9 and pn.grad.abs().max().item() <= 1e-3, k
            continue
        scale = max(pr.grad.abs().max().item(), 1e-3)
        assert (pn.grad.cpu().double() - pr.grad).abs().max().item() <= 5e-3 * scale, k
    assert (blk.norm.running_mean.cpu().double() - ref.norm.running_mean).abs().max().item() <= 1e-5
    assert (blk.norm.running_var.cpu().double() - ref.norm.running_var).abs().max().item() <= 1e-5
    assert int(blk.norm.num_batches_tracked) == 2
    if drop and stride == 1:
        assert blk.conv1.bias.grad.abs().max().item() > 1e-4        # does not cancel in front of a batch norm


def test_bn_upconcat_training_vs_fp64():
    """UpConcat(norm_op=BatchNorm3d) without attention, training mode: ConvTranspose3d + far zero pad + batch statistics
    that include the pad planes + LeakyReLU + concat, on the native kernels, against the torch modules in fp64."""
    import copy
    torch.manual_seed(6)
    up = network.UpConcat(16, 8, norm_op=torch.nn.BatchNorm3d).to(DEV).train()
    with torch.no_grad():
        up.conv_trans.up[2].weight.uniform_(0.5, 1.5)
        up.conv_trans.up[2].bias.uniform_(-0.3, 0.3)
    x = torch.randn(2, 16, 5, 6, 4, device=DEV, requires_grad=True)
    skip = torch.randn(2, 8, 10, 12, 8, device=DEV, requires_grad=True)
    g = torch.randn(2, 16, 10, 12, 8, device=DEV)
    ref = copy.deepcopy(up).cpu().double()
    xr = x.detach().cpu().double().requires_grad_(True)
    sr = skip.detach().cpu().double().requires_grad_(True)
    out_r = torch.cat((ref.conv_trans.up(xr), sr), dim=1)
    out_r.backward(g.cpu().double())
    calls, undo = _count_calls(ops.UpBNFn)
    try:
        out = up(x, skip)
    finally:
        undo()
    assert len(calls) == 1
    out.backward(g)
    assert (out.detach().cpu().double() - out_r.detach()).abs().max().item() <= 2e-4
    assert (x.grad.cpu().double() - xr.grad).abs().max().item() <= 2e-3 * max(1.0, xr.grad.abs().max().item())
    assert (skip.grad.cpu().double() - sr.grad).abs().max().item() <= 1e-6
    for (k, pn), (_, pr) in zip(up.named_parameters(), ref.named_parameters()):
        scale = max(pr.grad.abs().max().item(), 1e-3)
        assert (pn.grad.cpu().double() - pr.grad).abs().max().item() <= 5e-3 * scale, k
    bn, bnr = up.conv_trans.up[2], ref.conv_trans.up[2]
    assert (bn.running_mean.cpu().double() - bnr.running_mean).abs().max().item() <= 1e-5
    assert (bn.running_var.cpu().double() - bnr.running_var).abs().max().item() <= 1e-5


@pytest.mark.parametrize("feat,dtype", [(32, torch.bfloat16), (30, torch.bfloat16), (30, torch.float16)])
def test_bn_unet_training_16_bit_and_padded(feat, dtype):
    """A training step of the BatchNorm variant in 16-bit storage (F = 32 on the MFMA kernels, F = 30 channel-padded):
    logits close to the fp32 native run of the same weights and draws, finite gradients of the parameters' shapes,
    pad lanes stay out of gamma / beta / the running statistics."""
    torch.manual_seed(3)
    model = network.ResAttrBNUnet3D(2, feat, 1, 2).to(DEV).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout3d):
            m.p = 0.0
    x = torch.randn(2, 1, 32, 32, 32, device=DEV)
    y = (torch.rand(2, 32, 32, 32, device=DEV) > 0.6).long()
    import copy
    ref_model = copy.deepcopy(model)
    ref = ref_model(x)
    network.set_compute_dtype(model, dtype)
    logits = model(x)
    assert model.net._pad == (feat == 30)
    assert (logits - ref).abs().max().item() <= (0.25 if dtype == torch.bfloat16 else 0.05)
    L.HybirdLoss()(logits, y).backward()
    for k, p in model.named_parameters():
        if p.grad is None:          # the 1x1 skip conv of a block whose shapes match is constructed but unused
            assert "skip_conv" in k, k
            continue
        assert p.grad.shape == p.shape and torch.isfinite(p.grad).all(), k
    for (k, v), (_, vr) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        if "running_" in k:
            assert v.shape == vr.shape and (v - vr).abs().max().item() <= 0.05 * max(1.0, vr.abs().max().item()), k


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
    # Yardstick: the relative L2 distance to the fp32 run - stable to three digits across kernel families and seeds
    # (profiles/r04_attention_gate_yardstick.txt: 0.029-0.045 bf16, 0.0036-0.0057 fp16, the same values with the small
    # levels' norm kernels old or new) - plus a loose bound on the largest single deviation, which is one voxel's rounding
    # luck: 0.098-0.124 bf16 over three seeds and two kernel families, on either side of the 0.12 this test used to allow.
    d = (logits - ref).float()
    rel = (d.norm() / ref.float().norm()).item()
    assert rel <= (0.06 if dtype == torch.bfloat16 else 0.008), rel
    assert d.abs().max().item() <= (0.2 if dtype == torch.bfloat16 else 0.03)
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
def test_bn_inference_16_bit_and_padded(feat, dtype, monkeypatch):
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
    out = model(x)                                          # training mode is native too: still padded
    assert model.net._pad == (feat == 30) and torch.isfinite(out).all()
    monkeypatch.setenv("RU3D_BN_TRAIN", "torch")
    out = model(x)                                          # the torch-module cross-check path: un-padded, fp32
    assert not model.net._pad and torch.isfinite(out).all()
