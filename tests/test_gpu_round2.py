"""Round-2 features on a real MI355X (run with `-m gpu`):
  * channel padding: F = 30 widths (the reference's default num_features, network.py:107; BASELINE config 4) on the
    MFMA kernels - padded weight packs, un-padded weight gradients, whole-net parity against the oracle's 16-bit
    storage model and against the un-padded generic-kernel run of the same model;
  * activation checkpointing (BASELINE config 5): gradients bit-equal to the plain run, less memory."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import _native as N  # noqa: E402
import _ops as ops  # noqa: E402
import loss as L  # noqa: E402
import network  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


def _zero_padded(w, cout_seg, cin_seg, transposed=False):
    """Reference construction of the padded weight on the host: segments of `seg` channels -> cpad(seg) each."""
    def pad_dim(t, dim, seg):
        if not seg:
            return t
        parts = []
        for s0 in range(0, t.shape[dim], seg):
            p = t.narrow(dim, s0, seg)
            padw = [0, 0] * (t.dim() - dim - 1) + [0, ops.cpad(seg) - seg]
            parts.append(torch.nn.functional.pad(p, padw))
        return torch.cat(parts, dim)
    od, idim = (1, 0) if transposed else (0, 1)
    return pad_dim(pad_dim(w, od, cout_seg), idim, cin_seg).contiguous()


@pytest.mark.parametrize("cout,cin,k,stride,cout_seg,cin_seg,conv_t", [
    (30, 30, 3, 1, 30, 30, False), (60, 30, 3, 2, 60, 30, False), (30, 60, 3, 1, 30, 30, False),
    (30, 60, 1, 1, 30, 30, False), (120, 60, 1, 2, 120, 60, False), (30, 1, 3, 1, 30, 0, False),
    (3, 30, 1, 1, 0, 30, False), (30, 60, 3, 2, 30, 60, True), (64, 30, 3, 1, 0, 30, False)])
def test_padded_pack_equals_pack_of_zero_padded_weight(cout, cin, k, stride, cout_seg, cin_seg, conv_t):
    """ru3d_pack_weights with channel segments == the plain pack of the explicitly zero-padded weight, every role."""
    g = torch.Generator().manual_seed(cout * 131 + cin)
    shape = (cin, cout, k, k, k) if conv_t else (cout, cin, k, k, k)
    w = torch.randn(shape, generator=g).to(DEV)
    wp = _zero_padded(w, cout_seg, cin_seg, conv_t)
    roles = (N.ROLE_CONVT_FWD, N.ROLE_CONVT_DGRAD) if conv_t else (N.ROLE_CONV_FWD, N.ROLE_CONV_DGRAD)
    for role in roles:
        a = ops.pack_weights([(w, role, stride, cout_seg, cin_seg)], torch.bfloat16)[0]
        b = ops.pack_weights([(wp, role, stride)], torch.bfloat16)[0]
        torch.cuda.synchronize()
        co_p, ci_p = (wp.shape[1], wp.shape[0]) if conv_t else (wp.shape[0], wp.shape[1])
        n = N.lib.ru3d_packed_weight_bytes(co_p, ci_p, k, 2 if conv_t else stride, role, N.BF16)   # packs are 256-B slots
        assert 0 < n <= a.numel() == b.numel() and torch.equal(a[:n], b[:n]), (role, n, a.numel(), b.numel())
    # the weight gradient path back: un-padding the padded tensor returns the original
    dw = wp.clone()
    if conv_t:
        back = ops.unpad_wgrad(dw, cin, cout, cin_seg, cout_seg)
    else:
        back = ops.unpad_wgrad(dw, cout, cin, cout_seg, cin_seg)
    assert torch.equal(back, w)


def test_padded_bias_pack():
    b = torch.arange(1, 61, dtype=torch.float32, device=DEV)
    w = torch.randn(60, 30, 1, 1, 1, device=DEV)
    packs = ops.pack_weights([(w, N.ROLE_CONV_FWD, 1, 30, 30), (b, N.ROLE_BIAS, 1, 30, 0)], torch.bfloat16)
    got = packs[1][:4 * 64].view(torch.float32).cpu()
    want = torch.zeros(64)
    want[0:30] = b[:30].cpu()
    want[32:62] = b[30:].cpu()
    assert torch.equal(got, want)


def _run(model, x, y, crit):
    model.zero_grad(set_to_none=True)
    logits = model(x)
    loss = crit(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().float().cpu(), float(loss.detach()), {k: p.grad.detach().cpu().clone()
                                                         for k, p in model.named_parameters() if p.grad is not None}


def test_f30_padded_mfma_path_vs_storage_model_and_generic_path(monkeypatch):
    """ResUnet3D(num_pool=2, num_features=30) in bf16: the padded MFMA path must (a) expose the reference's shapes
    (logits [N,3,...], gradients of the parameters' shapes), (b) sit as close to the float64 truth as the oracle's
    bf16 storage model (err_HIP <= 1.3 err_model + 0.01, as for F = 32), (c) agree with the un-padded generic-kernel
    run of the same model to bf16 rounding."""
    torch.manual_seed(3)
    model = network.ResUnet3D(num_pool=2, num_features=30, in_channels=1, out_channels=3).to(DEV).eval()
    network.set_compute_dtype(model, torch.bfloat16)
    assert model.net._pad and model.net.encode_blocks[0].res_blocks[0]._pad
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dims = (32, 32, 32)
    x = O.synth_image((2, 1) + dims, 21)
    y = O.phantom_labels(2, dims, 3)
    crit = L.HybirdLoss(weight_v=[1, 10, 20])
    logits, loss, grads = _run(model, x.to(DEV), y.to(DEV), crit)
    assert tuple(logits.shape) == (2, 3) + dims
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert p.grad.shape == p.shape and p.grad.dtype == torch.float32, k
    ref_loss, ref_logits, g64 = O.train_step({k: v.double() for k, v in w.items()}, x.double(), y, 2,
                                             loss_kwargs={"weight_v": [1, 10, 20]})
    O.set_storage(torch.bfloat16)
    try:
        _, lsim, gsim = O.train_step(w, x, y, 2, loss_kwargs={"weight_v": [1, 10, 20]})
    finally:
        O.set_storage(None)
    e_log_hip = (logits - ref_logits.float()).abs().max().item()
    e_log_sim = (lsim - ref_logits.float()).abs().max().item()
    assert e_log_hip <= 1.5 * e_log_sim + 0.02, (e_log_hip, e_log_sim)
    assert abs(loss - float(ref_loss)) <= 5e-3
    checked = 0
    for k, gr in grads.items():
        if not k.endswith("weight"):
            continue
        t = g64[k].float()
        e_hip = ((gr - t).norm() / t.norm()).item()
        e_sim = ((gsim[k] - t).norm() / t.norm()).item()
        assert e_hip <= 1.3 * e_sim + 0.01, "F=30 bf16 grad %s: HIP %.3f vs storage model %.3f" % (k, e_hip, e_sim)
        checked += 1
    assert checked >= 20
    # (c) the same model on the generic (un-padded) bf16 kernels
    monkeypatch.setenv("RU3D_PAD_CHANNELS", "0")
    network.set_compute_dtype(model, torch.bfloat16)
    assert not model.net._pad
    logits_g, loss_g, grads_g = _run(model, x.to(DEV), y.to(DEV), crit)
    monkeypatch.delenv("RU3D_PAD_CHANNELS")
    assert (logits - logits_g).abs().max().item() <= 2 * e_log_sim + 0.02
    assert abs(loss - loss_g) <= 5e-3
    assert grads.keys() == grads_g.keys()


def test_f30_train_mode_with_dropout_and_forced_masks():
    """Padded path in train mode: Dropout3d masks of the real channels are honoured, pad lanes stay zero."""
    torch.manual_seed(4)
    blk = network.ResBlock(30, 30).to(DEV).train()
    blk._pad = True
    keep = (torch.rand(2, 30) > 0.5).float()
    blk._forced_keep = keep
    xr = torch.randn(2, 30, 8, 8, 8)
    xp = torch.zeros(2, 32, 8, 8, 8)
    xp[:, :30] = xr
    out = blk(ops.as_input(xp.to(DEV), torch.bfloat16)).float().cpu()
    assert tuple(out.shape) == (2, 32, 8, 8, 8)
    assert float(out[:, 30:].abs().max()) == 0.0
    # reference: the un-padded generic path with the same mask
    blk._pad = False
    ref = blk(ops.as_input(xr.to(DEV), torch.bfloat16)).float().cpu()
    assert (out[:, :30] - ref).abs().max().item() <= 0.08      # two bf16 kernels' rounding on O(3) activations
    blk._forced_keep = None
    blk._pad = True
    out2 = blk(ops.as_input(xp.to(DEV), torch.bfloat16))
    assert torch.isfinite(out2).all() and float(out2[:, 30:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype,feat", [(torch.float32, 8), (torch.bfloat16, 32), (torch.bfloat16, 30)])
def test_checkpointing_gradients_bit_equal_and_memory_drops(dtype, feat):
    torch.manual_seed(5)
    model = network.ResUnet3D(num_pool=2, num_features=feat, in_channels=1, out_channels=3).to(DEV).train()
    network.set_compute_dtype(model, dtype)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout3d):
            m.p = 0.0
    dims = (64, 64, 64) if feat >= 30 else (32, 32, 32)
    x = O.synth_image((1, 1) + dims, 31).to(DEV)
    y = O.phantom_labels(1, dims, 3).to(DEV)
    crit = L.HybirdLoss()

    def measure():
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = crit(logits, y)
        held = torch.cuda.memory_allocated() - base          # activations kept for backward
        loss.backward()
        torch.cuda.synchronize()
        return (logits.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()
                                          if p.grad is not None}, held)

    l0, g0, held0 = measure()
    network.set_checkpointing(model, True)
    l1, g1, held1 = measure()
    network.set_checkpointing(model, False)
    assert torch.equal(l0, l1)
    assert g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert held1 < 0.62 * held0, (held0, held1)


@pytest.mark.parametrize("cout,cin,k,stride,conv_t", [(32, 32, 3, 1, False), (64, 32, 3, 2, False), (128, 64, 3, 1, False),
                                                      (32, 64, 1, 1, False), (64, 128, 3, 2, True), (96, 32, 3, 1, False)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_two_role_pack_equals_single_role_packs(cout, cin, k, stride, conv_t, dtype):
    """ru3d_pack_weights packs the forward and the input-gradient form of a weight from one read of the source
    (pack_pair_kernel); each must equal what the one-weight entry point ru3d_pack_weight (element-wise gather kernel)
    produces for that role - also when only one of the two roles is requested."""
    g = torch.Generator().manual_seed(cout + 7 * cin + k)
    shape = (cin, cout, k, k, k) if conv_t else (cout, cin, k, k, k)
    w = torch.randn(shape, generator=g).to(DEV)
    roles = (N.ROLE_CONVT_FWD, N.ROLE_CONVT_DGRAD) if conv_t else (N.ROLE_CONV_FWD, N.ROLE_CONV_DGRAD)
    both = ops.pack_weights([(w, roles[0], stride), (w, roles[1], stride)], dtype)
    for role, got in zip(roles, both):
        ref = ops.pack_weight(w, role, dtype, stride)
        single = ops.pack_weights([(w, role, stride)], dtype)[0]
        torch.cuda.synchronize()
        n = ref.numel()
        assert torch.equal(got[:n], ref) and torch.equal(single[:n], ref), role


def test_mfma_packed_weight_is_not_handed_to_the_direct_kernel():
    """the pack layout is chosen from (channels, k, dtype); a conv of an MFMA-packed shape with a bf16 input
    and an fp32 output has no MFMA kernel and must fail instead of reading the pack as [tap][cin][cout_pad]."""
    x = ops.as_input(torch.randn(1, 32, 4, 4, 4, device=DEV), torch.bfloat16)
    w = torch.randn(32, 32, 3, 3, 3, device=DEV)
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    with pytest.raises(N.Ru3dError, match="MFMA fragment order"):
        ops.conv_fwd(x, pw, None, 32, 3, 1, out_dtype=torch.float32)
    y = ops.conv_fwd(x, pw, None, 32, 3, 1)                        # the supported combination still runs
    assert y.dtype == torch.bfloat16 and torch.isfinite(y).all()


@pytest.mark.parametrize("shape", [(2, 32, 16, 32, 64), (1, 32, 8, 8, 32), (2, 64, 8, 16, 32), (1, 16, 6, 10, 12),
                                   (6, 32, 16, 64, 128),      # columns > workgroups, samples change inside a workgroup
                                   (2, 32, 16, 64, 80),       # W = 2.5 columns of 32: the last column's far half idles
                                   (2, 64, 16, 64, 64), (3, 64, 8, 64, 96),     # the 64-channel sliding kernel's backward sums
                                   (2, 64, 16, 32, 40)])      # ... and their EDGE form (masked columns must not count)
def test_conv_dgrad_in_bwd_fused_equals_two_calls(shape):
    """ru3d_conv3d_dgrad_in_bwd (the IN + LeakyReLU backward sums taken in the epilogue of the sliding 32- / 64-channel conv)
    against ru3d_conv3d_dgrad followed by ru3d_in_lrelu_bwd: same da up to the rounding of the sums (fp32 partials per
    wave instead of double partials per block), and the plain two-call path on shapes the fused form does not take."""
    n, c, d, h, w = shape
    torch.manual_seed(5)
    dt = torch.bfloat16
    y1 = ops.as_input(torch.randn(n, c, d, h, w, device=DEV) * 2 + 0.3, dt)
    mean, scale = ops.in_stats(y1)
    a1 = ops.in_lrelu_fwd(y1, mean, scale)
    dy2 = ops.as_input(torch.randn(n, c, d, h, w, device=DEV), dt)
    wt = torch.randn(c, c, 3, 3, 3, device=DEV) * (1.0 / (27 * c) ** 0.5)
    pwd = ops.pack_weight(wt, N.ROLE_CONV_DGRAD, dt, 1)
    da = ops.conv_dgrad(dy2, pwd, tuple(a1.shape), 3, 1)
    ref, _ = ops.in_lrelu_bwd(da, a1, y1, mean, scale)
    got = ops.conv_dgrad_in_bwd(dy2, pwd, a1, mean, scale)
    torch.cuda.synchronize()
    r, g = ref.float(), got.float()
    assert torch.isfinite(g).all()
    err = (g - r).abs().max().item()
    # one bf16 ulp of the largest element at most (2^-7 relative at the bottom of its binade): the two paths differ in
    # the summation order of the backward's sums, nothing else
    assert err <= 2.0 ** -7 * r.abs().max().item() + 1e-6, (err, r.abs().max().item())
    # the bulk is bit-equal: only elements whose value sits on a bf16 rounding boundary may move by one ulp
    assert (g != r).float().mean().item() < 0.02
