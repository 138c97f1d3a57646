"""fp16 storage (RU3D_F16: the reference's apex-O1 arithmetic, trainer.py:492-493, 538-542; BASELINE config 4) on a
real MI355X: the second build of the kernel sources inside libru3d.so (IEEE half elements, v_mfma_f32_32x32x16_f16,
fp32 accumulation) against torch on fp16-rounded operands, the whole net against the oracle's fp16 storage model, and
the dynamic loss scaler.  Run with `-m gpu`."""
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
import optim  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")
H = torch.float16
EPS = 2.0 ** -11      # half an ulp of fp16 relative to the value


def _close(a, b, rtol, atol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * max(b.abs().max().item(), 1e-30)
    assert err <= lim, "%s: max err %.3e > %.3e" % (what, err, lim)


@pytest.mark.parametrize("shape", [(2, 32, 32, 5, 9, 37), (1, 64, 64, 3, 10, 12), (1, 128, 64, 5, 5, 7),
                                   (2, 32, 32, 64, 64, 64), (2, 32, 64, 32, 64, 64), (1, 64, 64, 16, 32, 32), (2, 64, 64, 32, 32, 64), (2, 64, 32, 16, 32, 64),
                                   (1, 256, 256, 8, 8, 8)])
def test_conv3_s1_fp16(shape):
    """3x3x3 stride-1 conv (halo-tile, producer/consumer, D-sliding and split-K kernels) with bias + residual, its
    input gradient and weight gradient in fp16 storage: torch CPU fp32 conv on the fp16-rounded operands."""
    n, cin, cout, d, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    xv = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, cout, d, h, w, generator=g)
    x = ops.as_input(xv.to(DEV), H)
    res = ops.as_input(r.to(DEV), H)
    assert x.dtype == H
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, H, 1)
    y = ops.conv_fwd(x, pw, b.to(DEV), cout, 3, 1, res=res)
    xr = xv.half().float().requires_grad_(True)
    wr = wt.half().float().requires_grad_(True)
    ref = torch.nn.functional.conv3d(xr, wr, b, padding=1)
    _close(y, ref + r.half().float(), 1.5 * EPS, 1e-3, "fp16 conv fwd %s" % (shape,))
    gy = torch.randn(n, cout, d, h, w, generator=g)
    gyd = ops.as_input(gy.to(DEV), H)
    pwd = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, H, 1)
    gx = ops.conv_dgrad(gyd, pwd, (n, cin, d, h, w), 3, 1)
    ref.backward(gy.half().float())
    _close(gx, xr.grad, EPS, 1e-3, "fp16 conv dgrad %s" % (shape,))
    gw = ops.conv_wgrad(x, gyd, 3, 1)
    _close(gw, wr.grad, 5e-4, 1e-3, "fp16 wgrad %s" % (shape,))
    # fused InstanceNorm statistics of the conv output
    y2, mean, scale = ops.conv_fwd_in(x, pw, b.to(DEV), cout, 3, 1)
    yf = y2.float()
    _close(mean.reshape(n, cout), yf.mean(dim=(2, 3, 4)), 0, 2e-4, "fp16 fused mean")
    var = yf.var(dim=(2, 3, 4), unbiased=False)
    _close(scale.reshape(n, cout), 1.0 / torch.sqrt(var + 1e-5), 2e-3, 0, "fp16 fused scale")


@pytest.mark.parametrize("cin,cout,k,stride,dims", [(32, 64, 3, 2, (8, 8, 16)), (64, 32, 1, 1, (5, 6, 7)),
                                                    (32, 64, 1, 2, (7, 8, 9)), (32, 64, 3, 2, (31, 32, 66)),
                                                    (64, 32, 1, 1, (40, 48, 56))])
def test_direct_forms_fp16(cin, cout, k, stride, dims):
    g = torch.Generator().manual_seed(cin + cout + k + stride)
    d, h, w = dims
    xv = torch.randn(2, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, k, generator=g) * (1.0 / (k ** 3 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    x = ops.as_input(xv.to(DEV), H)
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, H, stride)
    y = ops.conv_fwd(x, pw, b.to(DEV), cout, k, stride)
    xr = xv.half().float().requires_grad_(True)
    wr = wt.half().float().requires_grad_(True)
    ref = torch.nn.functional.conv3d(xr, wr, b, stride=stride, padding=k // 2)
    _close(y, ref, EPS, 1e-3, "fp16 direct fwd")
    gy = torch.randn(ref.shape, generator=g)
    gyd = ops.as_input(gy.to(DEV), H)
    pwd = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, H, stride)
    gx = ops.conv_dgrad(gyd, pwd, tuple(xv.shape), k, stride)
    ref.backward(gy.half().float())
    _close(gx, xr.grad, EPS, 2e-3, "fp16 direct dgrad")
    gw = ops.conv_wgrad(x, gyd, k, stride)
    _close(gw, wr.grad, 5e-4, 1e-3, "fp16 direct wgrad")


@pytest.mark.parametrize("cin,cout,dims", [(64, 32, (3, 4, 5)), (64, 32, (16, 15, 33)), (32, 32, (9, 16, 64))])
def test_convtranspose_fp16(cin, cout, dims):
    g = torch.Generator().manual_seed(cin + cout)
    d, h, w = dims
    xv = torch.randn(2, cin, d, h, w, generator=g)
    wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (27 * cin / 8) ** 0.5)
    b = torch.randn(cout, generator=g)
    x = ops.as_input(xv.to(DEV), H)
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_FWD, H)
    y = ops.convt_fwd(x, pw, b.to(DEV), cout)
    xr = xv.half().float().requires_grad_(True)
    wr = wt.half().float().requires_grad_(True)
    ref = torch.nn.functional.pad(torch.nn.functional.conv_transpose3d(xr, wr, b, stride=2, padding=1),
                                  (0, 1, 0, 1, 0, 1))
    _close(y, ref, EPS, 1e-3, "fp16 convT fwd")
    gy = torch.randn(ref.shape, generator=g)
    gy[:, :, -1] = 0
    gy[:, :, :, -1] = 0
    gy[:, :, :, :, -1] = 0
    gyd = ops.as_input(gy.to(DEV), H)
    pwd = ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_DGRAD, H)
    gx = ops.convt_dgrad(gyd, pwd, tuple(xv.shape))
    ref.backward(gy.half().float())
    _close(gx, xr.grad, EPS, 2e-3, "fp16 convT dgrad")
    gw = ops.convt_wgrad(x, gyd)
    _close(gw, wr.grad, 5e-4, 1e-3, "fp16 convT wgrad")


def test_norm_kernels_and_stem_fp16():
    g = torch.Generator().manual_seed(8)
    yv = torch.randn(2, 32, 6, 10, 12, generator=g) * 2 + 0.5
    rv = torch.randn(2, 32, 6, 10, 12, generator=g)
    y = ops.as_input(yv.to(DEV), H)
    res = ops.as_input(rv.to(DEV), H)
    mean, scale = ops.in_stats(y)
    yr = yv.half().float()
    rr = rv.half().float()
    xhat = (yr - yr.mean(dim=(2, 3, 4), keepdim=True)) / torch.sqrt(yr.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
                                                                    + 1e-5)
    out = ops.in_lrelu_fwd(y, mean, scale, res=res)
    _close(out, torch.nn.functional.leaky_relu(xhat + rr, 0.01), EPS, 1e-3, "fp16 IN apply")
    # backward against autograd on the same fp16-rounded inputs
    yr.requires_grad_(True)
    rr.requires_grad_(True)
    o = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(yr, eps=1e-5) + rr, 0.01)
    go = torch.randn(o.shape, generator=g)
    o.backward(go.half().float())
    dy, gpre = ops.in_lrelu_bwd(ops.as_input(go.to(DEV), H), out, y, mean, scale, want_gpre=True)
    _close(gpre, rr.grad, EPS, 1e-3, "fp16 IN bwd gpre")
    _close(dy, yr.grad, 4 * EPS, 2e-3, "fp16 IN bwd dy")
    # stem 1 -> 32
    xv = torch.randn(2, 1, 8, 16, 64, generator=g)
    wt = torch.randn(32, 1, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(32, generator=g)
    xs = ops.as_input(xv.to(DEV), H)
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, H, 1)
    ys = ops.conv_fwd(xs, pw, b.to(DEV), 32, 3, 1)
    refs = torch.nn.functional.conv3d(xv.half().float(), wt.half().float(), b, padding=1)
    _close(ys, refs, EPS, 1e-3, "fp16 stem")


@pytest.mark.parametrize("feat", [8, 32, 30])
def test_whole_net_fp16_vs_storage_model(feat):
    """ResUnet3D(2, F, 1, 3) in fp16 storage: F = 8 runs the generic kernels, 32 the MFMA kernels, 30 the padded MFMA
    path.  Yardstick as for bf16: the oracle's storage model (every inter-kernel tensor and weight rounded to fp16,
    exact arithmetic inside ops); err_HIP <= 1.3 err_model + 0.01 on the weight gradients, logits within
    1.5 err_model + 2e-3, argmax flips only inside the margin band."""
    torch.manual_seed(9)
    model = network.ResUnet3D(2, feat, 1, 3).to(DEV).eval()
    network.set_compute_dtype(model, H)
    assert model.net._pad == (feat == 30)
    w = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dims = (32, 32, 32)
    x = O.synth_image((1, 1) + dims, 17)
    y = O.phantom_labels(1, dims, 3)
    crit = L.HybirdLoss(weight_v=[1, 10, 20])
    logits = model(x.to(DEV))
    assert logits.dtype == torch.float32
    loss = crit(logits, y.to(DEV))
    loss.backward()
    ref_loss, ref_logits, g64 = O.train_step({k: v.double() for k, v in w.items()}, x.double(), y, 2,
                                             loss_kwargs={"weight_v": [1, 10, 20]})
    O.set_storage(H)
    try:
        _, lsim, gsim = O.train_step(w, x, y, 2, loss_kwargs={"weight_v": [1, 10, 20]})
    finally:
        O.set_storage(None)
    got = logits.detach().cpu()
    ref = ref_logits.float()
    e_hip = (got - ref).abs().max().item()
    e_sim = (lsim - ref).abs().max().item()
    assert e_hip <= 1.5 * e_sim + 2e-3, (e_hip, e_sim)
    top2 = ref.topk(2, dim=1).values
    flips = got.argmax(1) != ref.argmax(1)
    assert not (flips & ((top2[:, 0] - top2[:, 1]) > 4 * e_sim + 2e-3)).any()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-3
    checked = 0
    for k, p in model.named_parameters():
        if p.grad is None or not k.endswith("weight"):
            continue
        t = g64[k].float()
        eh = ((p.grad.cpu() - t).norm() / t.norm()).item()
        es = ((gsim[k] - t).norm() / t.norm()).item()
        assert eh <= 1.3 * es + 0.01, "fp16 grad %s: HIP %.4f vs storage model %.4f" % (k, eh, es)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("fused", [True, False])
def test_loss_scaler_skips_on_overflow_and_grows(fused):
    torch.manual_seed(10)
    model = network.ResUnet3D(2, 8, 1, 2).to(DEV).eval()
    network.set_compute_dtype(model, H)
    opt = (optim.Adam if fused else torch.optim.Adam)(model.parameters(), lr=1e-4)
    x = O.synth_image((1, 1, 32, 32, 32), 3).to(DEV)
    y = O.phantom_labels(1, (32, 32, 32), 2).to(DEV)
    crit = L.HybirdLoss()
    sc = optim.LossScaler(init_scale=2.0 ** 40, growth_interval=2)      # 2^40 overflows fp16 activations' gradients
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt.zero_grad()
    sc.scale(crit(model(x), y)).backward()
    assert sc.step(opt) is False and sc.skipped_steps == 1 and sc.loss_scale == 2.0 ** 39
    for k, v in model.state_dict().items():
        assert torch.equal(v, w0[k]), k                                  # skipped: nothing moved
    assert len(opt.state) == 0 or all(float(s["step"]) == 0 for s in opt.state.values())
    sc.loss_scale = 1024.0
    sc._scale_t.fill_(1024.0)
    losses = []
    for i in range(4):
        opt.zero_grad()
        l = crit(model(x), y)
        sc.scale(l).backward()
        assert sc.step(opt) is True
        losses.append(float(l.detach()))
    assert sc.loss_scale == 4096.0 and sc.growth_tracker == 0          # doubled twice (interval 2)
    assert losses[-1] < losses[0]
    # the unscaled update equals the plain fp16 run without a scaler (power-of-two scales are exact until overflow)
    torch.manual_seed(10)
    ref_model = network.ResUnet3D(2, 8, 1, 2).to(DEV).eval()
    network.set_compute_dtype(ref_model, H)
    ropt = (optim.Adam if fused else torch.optim.Adam)(ref_model.parameters(), lr=1e-4)
    for i in range(4):
        ropt.zero_grad()
        crit(ref_model(x), y).backward()
        ropt.step()
    diffs = torch.cat([(a - b).abs().flatten() for a, b in zip(model.state_dict().values(),
                                                               ref_model.state_dict().values())])
    # close, not identical: without the scale, gradient terms at fp16's underflow level are lost (which is what the
    # scale is for), and Adam's normalised step (at most ~3.2 lr) then goes a slightly different way.  Measured: median
    # 1e-5 (a tenth of one lr step after four steps), max 5.8e-4
    assert float(diffs.median()) <= 5e-5 and float(diffs.max()) <= 4 * 3.2e-4, (float(diffs.median()), float(diffs.max()))
    st = sc.state_dict()
    sc2 = optim.LossScaler()
    sc2.load_state_dict(st)
    assert sc2.loss_scale == 4096.0


def test_fp16_step_replays_from_a_graph_with_the_scaler_on_the_device():
    """The reference's own training mode (apex O1: trainer.py:492-496, 538-542) as ONE captured hipGraph: the loss scaler's
    state lives on the device (ru3d_amp_state), the update kernel skips itself on overflow, ru3d_amp_update applies apex's
    schedule.  Trajectory against the eager LossScaler loop on the same batches: the same skips, the same scale, the same
    step counts; weights equal up to the bias corrections' last bit (pow on the device vs on the host)."""
    import graph

    def setup():
        torch.manual_seed(12)
        model = network.ResUnet3D(2, 8, 1, 2).to(DEV).train()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout3d):
                m.p = 0.0
        network.set_compute_dtype(model, H)
        return model, optim.Adam(model.parameters(), lr=1e-3), L.HybirdLoss()

    batches = [(O.synth_image((1, 1, 32, 32, 32), 40 + i).to(DEV), O.phantom_labels(1, (32, 32, 32), 2).to(DEV))
               for i in range(16)]
    # 2^30 overflows the fp16 gradients at first: a few skipped steps, halvings, then growth every 3 clean steps
    kw = dict(init_scale=2.0 ** 30, growth_interval=3)
    m_e, o_e, crit = setup()
    sc_e = optim.LossScaler(**kw)
    log_e = []
    for x, y in batches:
        o_e.zero_grad()
        sc_e.scale(crit(m_e(x), y)).backward()
        log_e.append(sc_e.step(o_e))
    m_g, o_g, crit_g = setup()
    sc_g = optim.LossScaler(**kw)
    step = graph.GraphedTrainStep(m_g, crit_g, o_g, warmup=2, scaler=sc_g)
    for x, y in batches:
        step(x, y)
    assert step.replays >= 3 and step.replays + step.eager_steps + step.skipped_warmup == len(batches)
    step.release()                                       # scaler and step counts come back to the host
    assert sc_g._dev is None
    assert log_e.count(False) >= 1                       # the scenario does contain skipped steps
    assert sc_g.loss_scale == sc_e.loss_scale and sc_g.skipped_steps == sc_e.skipped_steps
    assert sc_g.growth_tracker == sc_e.growth_tracker
    for p_e, p_g in zip(m_e.parameters(), m_g.parameters()):
        if p_e in o_e.state:
            assert float(o_e.state[p_e]["step"]) == float(o_g.state[p_g]["step"]) == log_e.count(True)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), k
    # and eager steps continue from there
    o_g.zero_grad()
    sc_g.scale(crit_g(m_g(batches[0][0]), batches[0][1])).backward()
    assert sc_g.step(o_g) in (True, False)
