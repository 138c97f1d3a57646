"""Full-size (BASELINE.json config 2: 2 x 128^3, F = 32, 4 levels) checks of the HIP path through
size-independent properties, plus one full-size comparison with the CPU oracle.  `-m gpu` only.

Properties: exact tap-count identities for conv / wgrad on constant inputs (every tile edge of the
128^3 grid is exercised), linearity under power-of-two scaling (exact in bf16), InstanceNorm moments,
loss checksums (class probabilities sum to N*V, one-hot counts are exact), determinism of the whole
training step (fixed-order reductions, no atomics)."""
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
S = 128


def _valid_counts(d, h, w):
    """number of in-range taps of a 3x3x3 / pad-1 stencil at every voxel"""
    def axis(n):
        v = torch.full((n,), 3.0)
        v[0] = v[-1] = 2.0
        return v
    return axis(d)[:, None, None] * axis(h)[None, :, None] * axis(w)[None, None, :]


def test_conv_and_wgrad_tap_count_identities_full_grid():
    n, c = 2, 32
    x = N.new_act(n, c, S, S, S, torch.bfloat16, DEV)
    x.fill_(1.0)
    w = torch.full((c, c, 3, 3, 3), 1.0 / c, device=DEV)
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    y = ops.conv_fwd(x, pw, None, c, 3, 1)                      # MFMA LDS-halo kernel
    counts = _valid_counts(S, S, S).to(DEV)
    assert torch.equal(y.float(), counts[None, None].expand(n, c, S, S, S))
    # wgrad with x = dy = 1: dW[co][ci][tap] = N * prod_axis (S - |k - 1|), exact in fp32
    gw = ops.conv_wgrad(x, x, 3, 1)
    ax = torch.tensor([S - 1.0, S, S - 1.0])
    expect = n * ax[:, None, None] * ax[None, :, None] * ax[None, None, :]
    assert torch.equal(gw.cpu(), expect[None, None].expand(c, c, 3, 3, 3))
    # stride-2 pooling conv (direct MFMA kernel) and its transposed-form input gradient
    w2 = torch.full((2 * c, c, 3, 3, 3), 1.0 / c, device=DEV)
    pw2 = ops.pack_weight(w2, N.ROLE_CONV_FWD, torch.bfloat16, 2)
    y2 = ops.conv_fwd(x, pw2, None, 2 * c, 3, 2)
    assert tuple(y2.shape) == (n, 2 * c, S // 2, S // 2, S // 2)
    def axis2(m):
        v = torch.full((m,), 3.0)
        v[0] = 2.0            # output o reads inputs 2o-1..2o+1; only o = 0 loses a tap (2*63+1 = 127 is valid)
        return v
    c2 = axis2(S // 2)[:, None, None] * axis2(S // 2)[None, :, None] * axis2(S // 2)[None, None, :]
    assert torch.equal(y2.float(), c2.to(DEV)[None, None].expand(n, 2 * c, S // 2, S // 2, S // 2))


def test_linearity_power_of_two_scaling_is_exact():
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((2, S, S, S, 32), generator=g, device=DEV).bfloat16().permute(0, 4, 1, 2, 3)
    w = torch.randn((32, 32, 3, 3, 3), generator=g, device=DEV) * 0.05
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    y1 = ops.conv_fwd(x, pw, None, 32, 3, 1)
    y4 = ops.conv_fwd(x * 4, pw, None, 32, 3, 1)
    assert torch.equal(y4, y1 * 4)
    assert torch.isfinite(y1).all()


def test_instance_norm_moments_full_size():
    g = torch.Generator(device=DEV).manual_seed(6)
    y = (torch.randn((2, S, S, S, 32), generator=g, device=DEV) * 3 + 1.5).bfloat16().permute(0, 4, 1, 2, 3)
    mean, scale = ops.in_stats(y)
    out = ops.in_lrelu_fwd(y, mean, scale).float()
    xhat = torch.where(out > 0, out, out / ops.LRELU_SLOPE)
    m = xhat.mean(dim=(2, 3, 4))
    v = xhat.var(dim=(2, 3, 4), unbiased=False)
    assert m.abs().max().item() < 5e-3 and (v - 1).abs().max().item() < 2e-2
    ref_mean = y.float().mean(dim=(2, 3, 4)).reshape(-1)
    assert (mean - ref_mean).abs().max().item() < 1e-4


def test_loss_checksums_full_size():
    g = torch.Generator(device=DEV).manual_seed(7)
    logits = torch.randn((2, S, S, S, 3), generator=g, device=DEV).permute(0, 4, 1, 2, 3).requires_grad_(True)
    y = torch.randint(0, 3, (2, S, S, S), generator=g, device=DEV)
    fn = L.HybirdLoss(weight_v=[1, 10, 20])
    v = fn(logits, y)
    assert torch.isfinite(v) and v.item() > 0
    v.backward()
    gsum = logits.grad.sum(dim=1)           # softmax Jacobian: the class gradients of every voxel sum to 0
    assert gsum.abs().max().item() < 1e-9
    # perfect prediction: dice -> 1, focal -> 0
    onehot = torch.nn.functional.one_hot(y, 3).permute(0, 4, 1, 2, 3).float() * 40.0
    assert L.HybirdLoss(weight_v=[1, 10, 20])(onehot, y).item() < 1e-5
    assert abs(L.Dice()(onehot, y).item() - 1.0) < 1e-6
    # uint8 labels give the same value as int64
    assert L.HybirdLoss(weight_v=[1, 10, 20])(logits.detach(), y.to(torch.uint8)).item() == pytest.approx(v.item(), abs=0)


def _config2(dtype, seed=0):
    torch.manual_seed(seed)
    m = network.ResUnet3D(4, 32, 1, 3).to(DEV)
    network.set_compute_dtype(m, dtype)
    return m


def test_training_step_is_deterministic_bf16():
    model = _config2(torch.bfloat16).eval()
    x = O.synth_image((2, 1, S, S, S), 1234).to(DEV)
    y = torch.randint(0, 3, (2, S, S, S), generator=torch.Generator().manual_seed(1)).to(DEV)
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
        loss.backward()
        runs.append((logits.detach().clone(), loss.item(),
                     {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k
    assert len(runs[0][2]) == 66 and all(torch.isfinite(g).all() for g in runs[0][2].values())


def test_config2_forward_fp32_vs_cpu_oracle_full_patch():
    """ResUnet3D(4,32,1,3) on a full 128^3 patch (N = 1 to keep the CPU side ~10 s), fp32 parity mode vs the
    CPU oracle on identical weights: logits within 5e-5 abs (measured 3e-6 - 7e-6 across boxes), argmax may differ
    only where the oracle's top-2 margin is below 5e-5 (measured: 10 flipped voxels of 2,097,152 with these untrained
    weights, 0 - 1 with the bench's), at most 32 flips,
    per-class Dice of the masks > 0.9999."""
    model = _config2(torch.float32).eval()
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = O.synth_image((1, 1, S, S, S), 4321)
    with torch.no_grad():
        got = model(x.to(DEV)).cpu()
        torch.set_num_threads(min(16, torch.get_num_threads()))
        ref = O.unet_forward(x, w, 4)
    assert (got - ref).abs().max().item() < 5e-5
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    flips = got.argmax(1) != ref.argmax(1)
    assert not (flips & (margin > 5e-5)).any()
    assert int(flips.sum()) <= 32
    for c in range(3):
        a, b = (got.argmax(1) == c).float(), (ref.argmax(1) == c).long()
        if b.sum() > 0:
            assert O.tversky(a, b).item() > 0.9999


# --------------------------------------------------------------------------- BASELINE config 5 (F = 64, P = 5, 192^3)
S5 = 192


def test_config5_tap_count_identities_192_cubed_64_channels():
    """64 -> 64 channels on the 1 x 192^3 grid (453 M elements per tensor: every index product of the kernels is
    exercised close to 2^31), and the 2048-channel bottleneck on its 6^3 grid."""
    n, c = 1, 64
    x = N.new_act(n, c, S5, S5, S5, torch.bfloat16, DEV)
    x.fill_(1.0)
    w = torch.full((c, c, 3, 3, 3), 1.0 / c, device=DEV)
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    y = ops.conv_fwd(x, pw, None, c, 3, 1)
    counts = _valid_counts(S5, S5, S5).to(DEV)
    assert torch.equal(y.float(), counts[None, None].expand(n, c, S5, S5, S5))
    del y
    gw = ops.conv_wgrad(x, x, 3, 1)
    ax = torch.tensor([S5 - 1.0, S5, S5 - 1.0])
    expect = n * ax[:, None, None] * ax[None, :, None] * ax[None, None, :]
    assert torch.equal(gw.cpu(), expect[None, None].expand(c, c, 3, 3, 3))
    del x, gw
    # bottleneck: 2048 -> 2048 on 6^3 (sums of 2048 * 27 terms of 2^-11 are exact in fp32)
    cb, sb = 2048, 6
    xb = N.new_act(1, cb, sb, sb, sb, torch.bfloat16, DEV)
    xb.fill_(1.0)
    wb = torch.full((cb, cb, 3, 3, 3), 1.0 / cb, device=DEV)
    pwb = ops.pack_weight(wb, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    yb = ops.conv_fwd(xb, pwb, None, cb, 3, 1)
    cnt = _valid_counts(sb, sb, sb).to(DEV)
    assert torch.equal(yb.float(), cnt[None, None].expand(1, cb, sb, sb, sb))
    gwb = ops.conv_wgrad(xb, xb, 3, 1)
    axb = torch.tensor([sb - 1.0, sb, sb - 1.0])
    eb = axb[:, None, None] * axb[None, :, None] * axb[None, None, :]
    assert torch.equal(gwb[:64, :64].cpu(), eb[None, None].expand(64, 64, 3, 3, 3))
    assert torch.equal(gwb[-1, -1].cpu(), eb)


def test_config5_training_step_deterministic_and_checkpoint_bit_equal():
    """ResUnet3D(5, 64, 1, 3) on 1 x 192^3 in bf16 (1.86 G parameters): two runs give the same bits, and the
    checkpointed run (ResBlock interiors recomputed in backward) gives the same bits again with less memory held."""
    torch.manual_seed(0)
    model = network.ResUnet3D(5, 64, 1, 3).to(DEV).eval()
    network.set_compute_dtype(model, torch.bfloat16)
    x = O.synth_image((1, 1, S5, S5, S5), 99).to(DEV)
    y = torch.randint(0, 3, (1, S5, S5, S5), generator=torch.Generator().manual_seed(2)).to(DEV)
    crit = L.HybirdLoss(weight_v=[1, 10, 20])
    keys = ["net.conv.weight", "net.encode_blocks.0.res_blocks.0.conv1.weight", "net.pool_blocks.2.conv1.weight",
            "net.encode_blocks.5.res_blocks.4.conv2.weight", "net.up_blocks.0.conv_trans.up.0.weight",
            "net.decode_blocks.0.conv1.weight", "net.fc.weight"]
    params = dict(model.named_parameters())

    def run():
        model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        logits = model(x)
        loss = crit(logits, y)
        held = torch.cuda.memory_allocated() - base
        loss.backward()
        torch.cuda.synchronize()
        out = (logits.detach().clone(), float(loss.detach()), {k: params[k].grad.clone() for k in keys}, held)
        assert all(torch.isfinite(g).all() for g in out[2].values())
        return out

    a = run()
    b = run()
    network.set_checkpointing(model, True)
    c = run()
    for other in (b, c):
        assert torch.equal(a[0], other[0]) and a[1] == other[1]
        for k in keys:
            assert torch.equal(a[2][k], other[2][k]), k
    assert c[3] < 0.62 * a[3], (a[3], c[3])


# --------------------------------------------------------------------------- BASELINE config 4 (F = 30 padded, fp16, 160 x 160 x 80)
D4 = (160, 160, 80)


def test_config4_tap_count_identities_non_cubic_fp16():
    """The non-cubic 2 x 160 x 160 x 80 grid of config 4 in fp16 storage, 32 (= 30 padded) channels: every tile edge of
    the sliding / stride-2 / transposed kernels on extents that are not powers of two (160 = 5 x 32, 80, 40 = 2.5 x 16)."""
    n, c = 2, 32
    H = torch.float16
    d, h, w = D4
    x = N.new_act(n, c, d, h, w, H, DEV)
    x.fill_(1.0)
    wt = torch.full((c, c, 3, 3, 3), 1.0 / c, device=DEV)
    y = ops.conv_fwd(x, ops.pack_weight(wt, N.ROLE_CONV_FWD, H, 1), None, c, 3, 1)
    assert torch.equal(y.float(), _valid_counts(d, h, w).to(DEV)[None, None].expand(n, c, d, h, w))
    del y
    gw = ops.conv_wgrad(x, x, 3, 1)
    ax = [torch.tensor([s - 1.0, s, s - 1.0]) for s in D4]
    expect = n * ax[0][:, None, None] * ax[1][None, :, None] * ax[2][None, None, :]
    assert torch.equal(gw.cpu(), expect[None, None].expand(c, c, 3, 3, 3))
    del gw
    # stride-2 pooling conv 32 -> 64 (LDS-DMA gather form: 40 = 2.5 tiles of 16 along W) ...
    w2 = torch.full((2 * c, c, 3, 3, 3), 1.0 / c, device=DEV)
    y2 = ops.conv_fwd(x, ops.pack_weight(w2, N.ROLE_CONV_FWD, H, 2), None, 2 * c, 3, 2)

    def axis2(m):
        v = torch.full((m,), 3.0)
        v[0] = 2.0
        return v
    c2 = axis2(d // 2)[:, None, None] * axis2(h // 2)[None, :, None] * axis2(w // 2)[None, None, :]
    assert torch.equal(y2.float(), c2.to(DEV)[None, None].expand(n, 2 * c, d // 2, h // 2, w // 2))
    # ... and its transposed-form input gradient: dx[i] = number of outputs whose window holds i
    gy = N.new_act(n, 2 * c, d // 2, h // 2, w // 2, H, DEV)
    gy.fill_(1.0)
    wd = torch.full((2 * c, c, 3, 3, 3), 1.0 / (2 * c), device=DEV)
    gx = ops.conv_dgrad(gy, ops.pack_weight(wd, N.ROLE_CONV_DGRAD, H, 2), (n, c, d, h, w), 3, 2)

    def axisT(m):          # voxel i is read by outputs o with 2o-1 <= i <= 2o+1: two for odd i < m-1, one for even i and i = m-1
        v = torch.ones(m)
        v[1::2] = 2.0
        v[m - 1] = 1.0
        return v
    ct = axisT(d)[:, None, None] * axisT(h)[None, :, None] * axisT(w)[None, None, :]
    assert torch.equal(gx.float(), ct.to(DEV)[None, None].expand(n, c, d, h, w))
    # ConvTranspose 64 -> 32 from the half-resolution grid: the far planes are zero, the rest counts its taps
    xt = N.new_act(n, 2 * c, d // 2, h // 2, w // 2, H, DEV)
    xt.fill_(1.0)
    wtt = torch.full((2 * c, c, 3, 3, 3), 1.0 / (2 * c), device=DEV)
    yt = ops.convt_fwd(xt, ops.pack_weight(wtt, N.ROLE_CONVT_FWD, H), None, c)

    def axisU(m):          # output o of the (2m-1)-long transposed conv: one tap for even o, two for odd o; o = 2m-1 is the pad
        v = torch.ones(m)
        v[1::2] = 2.0
        v[m - 1] = 0.0
        return v
    cu = axisU(d)[:, None, None] * axisU(h)[None, :, None] * axisU(w)[None, None, :]
    assert torch.equal(yt.float(), cu.to(DEV)[None, None].expand(n, c, d, h, w))


def test_config4_step_deterministic_and_forward_vs_oracle_fp16():
    """ResUnet3D(4, 30, 1, 3) (85 M parameters, channels padded 30 -> 32) in fp16 storage on config 4's 160 x 160 x 80
    patches: two training steps at bs = 2 give the same bits; a bs = 1 forward stays within the fp16 storage model's
    distance of the fp32 CPU oracle (err_HIP <= 1.5 err_model + 2e-3, argmax flips only inside the margin band, per-class
    Dice of the masks > 0.99)."""
    H = torch.float16
    torch.manual_seed(0)
    model = network.ResUnet3D(4, 30, 1, 3).to(DEV).eval()
    network.set_compute_dtype(model, H)
    assert model.net._pad
    x = O.synth_image((2, 1) + D4, 44).to(DEV)
    y = torch.randint(0, 3, (2,) + D4, generator=torch.Generator().manual_seed(4)).to(DEV)
    crit = L.HybirdLoss(weight_v=[1, 148, 191], alpha=0.9, beta=0.1)      # nb_train_iib.py:19
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        logits = model(x)
        loss = crit(logits, y) * 1024.0          # a loss scale that keeps the fp16 gradients in range
        loss.backward()
        runs.append((logits.detach().clone(), float(loss.detach()),
                     {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k
        assert torch.isfinite(runs[0][2][k]).all(), k
    assert len(runs[0][2]) == 66
    del runs
    model.zero_grad(set_to_none=True)
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x1 = O.synth_image((1, 1) + D4, 45)
    with torch.no_grad():
        got = model(x1.to(DEV)).cpu()
        torch.set_num_threads(min(16, torch.get_num_threads()))
        ref = O.unet_forward(x1, w, 4)
        O.set_storage(H)
        try:
            sim = O.unet_forward(x1, w, 4)
        finally:
            O.set_storage(None)
    e_hip, e_sim = (got - ref).abs().max().item(), (sim - ref).abs().max().item()
    assert e_hip <= 1.5 * e_sim + 2e-3, (e_hip, e_sim)
    top2 = ref.topk(2, dim=1).values
    flips = got.argmax(1) != ref.argmax(1)
    assert not (flips & ((top2[:, 0] - top2[:, 1]) > 4 * e_sim + 2e-3)).any()
    for c in range(3):
        a, b = (got.argmax(1) == c).float(), (ref.argmax(1) == c).long()
        if b.sum() > 0:
            assert O.tversky(a, b).item() > 0.99
