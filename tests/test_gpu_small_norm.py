"""The whole-instance InstanceNorm + LeakyReLU kernels of the small levels (csrc/norm_small.hip; reference network.py:
384-386, 411-416 at 16^3 / 8^3 voxels): one launch for statistics + finalize + apply, forward and backward, optionally
straight from the split-K slices of the deepest convs.  Checked against float64 restatements of the reference formulas on
the STORED 16-bit tensors (so the only differences are the kernels' own fp32 arithmetic and the output rounding), and
bit for bit against the three-launch path of norm.hip where the contract demands it (a checkpointed block recomputes its
activation with ru3d_in_lrelu_fwd).  Run with `-m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import _native as N  # noqa: E402
import _ops as ops  # noqa: E402

DEV = torch.device("cuda:0")
F = torch.nn.functional
DTYPES = [torch.bfloat16, torch.float16]
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
SLOPE = 0.01

# (n, cin, cout, dims): 16^3 / 8^3 of config 2, config 4's 20x20x10 and 10x10x5 (ragged: masked pieces), a tiny one
SHAPES = [(2, 256, 256, (16, 16, 16)), (2, 512, 512, (8, 8, 8)), (2, 256, 256, (20, 20, 10)), (3, 512, 512, (10, 10, 5)),
          (1, 64, 32, (4, 6, 8)), (2, 128, 256, (12, 12, 12))]


def _rt(t, dt):
    return t.to(dt).float()


def _close(a, b, rtol, atol, what):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * max(b.abs().max().item(), 1e-30)
    assert err <= lim, "%s: max err %.3e > %.3e" % (what, err, lim)


def _stats64(y, drop):
    """mean, scale = s / sqrt(s^2 var + eps) per (n, c) of the stored tensor (network.py:384 with Dropout3d folded in)"""
    yd = y.double().cpu()
    n, c = yd.shape[:2]
    m = yd.mean(dim=(2, 3, 4))
    v = yd.var(dim=(2, 3, 4), unbiased=False)
    s = drop.double().cpu().view(n, c) if drop is not None else torch.ones(n, c, dtype=torch.float64)
    return m, s / (s * s * v + 1e-5).sqrt()


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,dims", SHAPES)
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_fwd_in_act_small(dt, n, cin, cout, dims, with_res):
    g = torch.Generator().manual_seed(n + cin + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    drop = ((torch.rand(n * cout, generator=g) > 0.5).float() * 2.0).to(DEV)
    rv = torch.randn(n, cout, d, h, w, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    res = ops.as_input(rv.to(DEV), dt) if with_res else None
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1)
    y, mean, scale, act = ops.conv_fwd_in_act(x, pw, b.to(DEV), cout, 3, 1, drop, res=res)
    # the conv itself (incl. the split-K slice sum inside the norm kernel at 512 channels): one output rounding
    ref = F.conv3d(_rt(xv, dt), _rt(wt, dt), b, padding=1)
    _close(y, ref, EPS[dt], 1e-3, "conv")
    # ... and bit for bit what the plain entry point stores (same slices, same summation order)
    assert torch.equal(y, ops.conv_fwd(x, pw, b.to(DEV), cout, 3, 1))
    m64, s64 = _stats64(y, drop)
    _close(mean.view(n, cout), m64, 0, 2e-6 * max(1.0, m64.abs().max().item()), "mean")
    _close(scale.view(n, cout), s64, 2e-6, 1e-7, "scale")
    yd = y.double().cpu()
    t = (yd - m64.view(n, cout, 1, 1, 1)) * s64.view(n, cout, 1, 1, 1)
    if with_res:
        t = t + res.double().cpu()
    _close(act, F.leaky_relu(t, SLOPE), EPS[dt], 1e-4, "act")
    # checkpoint contract: ru3d_in_lrelu_fwd on the same (y, mean, scale) gives the same bits
    assert torch.equal(act, ops.in_lrelu_fwd(y, mean, scale, res=res))


def _bwd64(gout, out, y, mean, scale, resid, zero_far):
    """float64 restatement of d/dy of out = lrelu(IN(y) (+ res)) on stored tensors (autograd of network.py:411-416)"""
    n, c = out.shape[:2]
    o = out.double().cpu()
    g = gout.double().cpu()
    gp = torch.where(o > 0, g, g * SLOPE)
    mu = mean.double().cpu().view(n, c, 1, 1, 1)
    sc = scale.double().cpu().view(n, c, 1, 1, 1)
    if resid:
        gp = gp.to(out.dtype).double()          # the stored pre-activation gradient is what both passes use
        xh = (y.double().cpu() - mu) * sc
    else:
        xh = torch.where(o > 0, o, o / SLOPE)
    m1 = gp.mean(dim=(2, 3, 4), keepdim=True)
    m2 = (gp * xh).mean(dim=(2, 3, 4), keepdim=True)
    dy = sc * (gp - m1 - xh * m2)
    if zero_far:
        dy[:, :, -1] = 0
        dy[:, :, :, -1] = 0
        dy[..., -1] = 0
    return dy, gp


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,c,dims", [(2, 256, (16, 16, 16)), (2, 512, (8, 8, 8)), (3, 96, (20, 20, 10)), (2, 512, (10, 10, 5)),
                                      (1, 8, (3, 5, 7))])
@pytest.mark.parametrize("resid", [False, True])
def test_in_lrelu_bwd_small(dt, n, c, dims, resid):
    g = torch.Generator().manual_seed(n + c + sum(dims) + int(resid))
    d, h, w = dims
    yv = torch.randn(n, c, d, h, w, generator=g) * 1.5 + 0.2
    rv = torch.randn(n, c, d, h, w, generator=g) * 0.5
    gv = torch.randn(n, c, d, h, w, generator=g)
    y = ops.as_input(yv.to(DEV), dt)
    m64, s64 = _stats64(y, None)
    mean = m64.float().reshape(-1).to(DEV)
    scale = s64.float().reshape(-1).to(DEV)
    res = ops.as_input(rv.to(DEV), dt) if resid else None
    out = ops.in_lrelu_fwd(y, mean, scale, res=res)
    gout = ops.as_input(gv.to(DEV), dt)
    zero_far = not resid
    if resid:
        dy, gpre, gsum = ops.in_lrelu_bwd(gout, out, y, mean, scale, want_gpre=True, want_gpre_sum=True)
    else:
        dy, gpre = ops.in_lrelu_bwd(gout, out, out, mean, scale, zero_far=True)
    ref_dy, ref_gp = _bwd64(gout, out, y, mean, scale, resid, zero_far)
    # dy carries its own output rounding plus the fp32 evaluation of a difference of O(1) terms
    _close(dy, ref_dy, EPS[dt], 2e-5 * float(s64.max()), "dy")
    if resid:
        assert torch.equal(gpre.double().cpu(), ref_gp), "stored pre-activation gradient"
        _close(gsum, ref_gp.sum(dim=(0, 2, 3, 4)), 1e-5, 1e-4, "sum of g' (skip conv bias gradient)")


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,c,dims", [(2, 512, (8, 8, 8)), (2, 256, (16, 16, 16)), (3, 512, (10, 10, 5))])
def test_conv_dgrad_in_bwd_small(dt, n, c, dims):
    """conv2's input gradient + the InstanceNorm / LeakyReLU backward of conv1's activation in one entry point: at 512
    channels the gradient exists only as split-K slices that the norm kernel sums itself."""
    g = torch.Generator().manual_seed(n + c + sum(dims))
    d, h, w = dims
    av = torch.randn(n, c, d, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, 3, generator=g) * (1.0 / (27 * c) ** 0.5)
    gv = torch.randn(n, c, d, h, w, generator=g)
    y = ops.as_input((av * 1.3 + 0.1).to(DEV), dt)
    m64, s64 = _stats64(y, None)
    mean = m64.float().reshape(-1).to(DEV)
    scale = s64.float().reshape(-1).to(DEV)
    act = ops.in_lrelu_fwd(y, mean, scale)
    dy2 = ops.as_input(gv.to(DEV), dt)
    pwd = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, dt, 1)
    got = ops.conv_dgrad_in_bwd(dy2, pwd, act, mean, scale)
    da = ops.conv_dgrad(dy2, pwd, tuple(act.shape), 3, 1)          # the stored form of the same gradient
    ref, _ = _bwd64(da, act, None, mean, scale, False, False)
    _close(got, ref, EPS[dt], 2e-5 * float(s64.max()), "dgrad + IN backward")


@pytest.mark.parametrize("dt", DTYPES + [torch.float32], ids=["bf16", "fp16", "f32"])
@pytest.mark.parametrize("n,c,dims", [(2, 32, (40, 36, 28)), (1, 64, (33, 20, 17)), (2, 128, (16, 16, 16)), (3, 24, (9, 10, 11)),
                                      (2, 512, (8, 8, 8))])
def test_in_lrelu_bwd_dy_sum(dt, n, c, dims):
    """The transposed conv's bias gradient (network.py:290-300: sum of dy over samples and voxels) out of the pass that
    writes dy, far planes zeroed as ConvTrans3D's padding wants: equal to the float64 sum of the STORED dy to the
    rounding of float partial sums, on the three-launch path (large tensors) and the whole-instance path (small ones)."""
    g = torch.Generator().manual_seed(n + c + sum(dims))
    d, h, w = dims
    yv = torch.randn(n, c, d, h, w, generator=g) * 1.5 + 0.2
    gv = torch.randn(n, c, d, h, w, generator=g)
    y = ops.as_input(yv.to(DEV), dt)
    m64, s64 = _stats64(y, None)
    mean = m64.float().reshape(-1).to(DEV)
    scale = s64.float().reshape(-1).to(DEV)
    out = ops.in_lrelu_fwd(y, mean, scale)
    gout = ops.as_input(gv.to(DEV), dt)
    dysum = torch.full((y.shape[1],), float("nan"), dtype=torch.float32, device=DEV)
    dy, _ = ops.in_lrelu_bwd(gout, out, out, mean, scale, zero_far=True, dy_sum=dysum)
    plain, _ = ops.in_lrelu_bwd(gout, out, out, mean, scale, zero_far=True)
    assert torch.equal(dy, plain), "dy itself does not depend on the extra output"
    ref = dy.double().sum(dim=(0, 2, 3, 4)).cpu()
    mag = dy.double().abs().sum(dim=(0, 2, 3, 4)).cpu()
    err = (dysum.double().cpu() - ref).abs()
    assert bool((err <= 2e-6 * mag + 1e-6).all()), float((err / (mag + 1e-30)).max())
