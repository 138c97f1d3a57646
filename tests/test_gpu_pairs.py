"""Tight per-op parity of the fused "pair" kernels that the default training path uses (reference network.py:403-416:
skip_conv + the block tail), through the C ABI on a real MI355X:

  ru3d_conv3d_s2_pair_fwd_in     pooling block: conv1 (k3 s2) + InstanceNorm statistics + skip_conv (k1 s2), one read of x
  ru3d_conv3d_s2_dgrad_pair      pooling block: both stride-2 input gradients (+ the parked concat share)
  ru3d_conv3d_s1_dgrad_pair      decoder block: conv1 (k3 s1) + skip_conv (k1 s1) input gradients (the "28th tap")
  ru3d_skip1x1_in_lrelu_fwd      decoder block tail: lrelu(IN(y2) + skip_conv(x)), the skip never stored

Reference: torch CPU fp32 conv on the operands rounded to the storage type; tolerance = one output rounding of the
storage type (2^-8 of the tensor's max for bf16, 2^-11 for fp16) - the same bar as test_mfma_conv_s1_bf16.  Both
16-bit builds, ragged extents (W = 40, 48, 66, 80: masked border tiles), residual on and off, and the benchmark's
2 x 128^3 shape for the two kernels that run there.  Run with `-m gpu`."""
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


def _close(a, b, rtol, atol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * max(b.abs().max().item(), 1e-30)
    assert err <= lim, "%s: max err %.3e > %.3e" % (what, err, lim)


def _rt(t, dt):
    """round to the storage type, back in fp32 (what the kernels read)"""
    return t.to(dt).float()


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,dims,with_drop", [(3, (32, 32, 64), True), (2, (31, 30, 66), True), (2, (16, 40, 80), False),
                                              (1, (64, 64, 64), False), (2, (24, 40, 48), True)])
def test_s2_pair_fwd_in(dt, n, dims, with_drop):
    """Pooling ResBlock forward (network.py:405-409): conv1 3x3x3 s2 with bias, its InstanceNorm statistics (with the
    Dropout3d factors folded into the scale) and skip_conv 1x1x1 s2 of the same input in one launch."""
    cin, cout = 32, 64
    g = torch.Generator().manual_seed(cin + cout + sum(dims) + n)
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    b3 = torch.randn(cout, generator=g)
    b1 = torch.randn(cout, generator=g)
    drop = ((torch.rand(n * cout, generator=g) > 0.5).float() * 2.0) if with_drop else None
    x = ops.as_input(xv.to(DEV), dt)
    out = ops.conv_s2_pair_fwd_in(x, ops.pack_weight(w3.to(DEV), N.ROLE_CONV_FWD, dt, 2), b3.to(DEV),
                                  ops.pack_weight(w1.to(DEV), N.ROLE_CONV_FWD, dt, 2), b1.to(DEV), cout,
                                  drop.to(DEV) if with_drop else None)
    assert out is not None, "no fused kernel for %s" % (dims,)
    y3, mean, scale, y1 = out
    r3 = F.conv3d(_rt(xv, dt), _rt(w3, dt), b3, stride=2, padding=1)
    r1 = F.conv3d(_rt(xv, dt), _rt(w1, dt), b1, stride=2)
    _close(y3, r3, EPS[dt], 1e-3, "pair fwd conv3 %s" % (dims,))
    _close(y1, r1, EPS[dt], 1e-3, "pair fwd conv1 %s" % (dims,))
    # statistics are those of the STORED tensor (what the apply pass reads)
    yf = y3.float().cpu().double()
    m = yf.mean(dim=(2, 3, 4)).reshape(-1)
    v = yf.var(dim=(2, 3, 4), unbiased=False).reshape(-1)
    sd = drop.double() if with_drop else torch.ones(n * cout, dtype=torch.float64)
    assert (mean.cpu().double() - m).abs().max().item() < 1e-5
    assert (scale.cpu().double() - sd / (sd * sd * v + 1e-5).sqrt()).abs().max().item() < 1e-4
    # the fused launch and the two separate entry points agree bit for bit (same tiles, same summation order)
    ya = ops.conv_fwd(x, ops.pack_weight(w3.to(DEV), N.ROLE_CONV_FWD, dt, 2), b3.to(DEV), cout, 3, 2)
    assert torch.equal(y3, ya)


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,dims,with_res", [(2, (32, 32, 64), True), (2, (31, 30, 66), True), (2, (31, 30, 66), False),
                                             (2, (16, 40, 80), True), (1, (64, 64, 64), False), (2, (24, 40, 48), True)])
def test_s2_dgrad_pair(dt, n, dims, with_res):
    """Pooling ResBlock backward: d/dx of conv1 (k3 s2) + d/dx of skip_conv (k1 s2) (+ the concat's parked share as the
    residual operand) in one launch."""
    cin, cout = 32, 64
    g = torch.Generator().manual_seed(cin + cout + sum(dims) + n)
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    xr = _rt(xv, dt).requires_grad_(True)
    y3 = F.conv3d(xr, _rt(w3, dt), None, stride=2, padding=1)
    y1 = F.conv3d(xr, _rt(w1, dt), None, stride=2)
    g3 = torch.randn(y3.shape, generator=g)
    g1 = torch.randn(y1.shape, generator=g)
    rv = torch.randn(xv.shape, generator=g)
    (y3 * _rt(g3, dt)).sum().backward(retain_graph=True)
    (y1 * _rt(g1, dt)).sum().backward()
    ref = xr.grad + (_rt(rv, dt) if with_res else 0.0)
    p3 = ops.pack_weight(w3.to(DEV), N.ROLE_CONV_DGRAD, dt, 2)
    p1 = ops.pack_weight(w1.to(DEV), N.ROLE_CONV_DGRAD, dt, 2)
    gx = ops.conv_s2_dgrad_pair(ops.as_input(g3.to(DEV), dt), p3, ops.as_input(g1.to(DEV), dt), p1, tuple(xv.shape),
                                res=ops.as_input(rv.to(DEV), dt) if with_res else None)
    assert gx is not None, "no fused kernel for %s" % (dims,)
    # sum of two convs (+ residual) rounded once; the residual form rounds conv sum and residual sum separately
    _close(gx, ref, (1.5 if with_res else 1.0) * EPS[dt], 1e-3, "s2 pair dgrad %s res=%s" % (dims, with_res))


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,n,dims", [(64, 32, 2, (16, 64, 64)), (32, 32, 3, (8, 64, 128)), (64, 32, 2, (32, 32, 64)),
                                             (64, 32, 2, (16, 64, 80)), (64, 32, 2, (32, 40, 48)),
                                             (64, 32, 2, (128, 128, 128))])
def test_s1_dgrad_pair(dt, cin, cout, n, dims):
    """Decoder ResBlock backward (network.py:403-416 with in != out): d/dx of conv1 (k3 s1) + d/dx of skip_conv (k1 s1)
    in one launch of the 32-channel sliding kernel (the 1x1 as a 28th tap).  The last case is the benchmark's own shape
    (2 x 128^3, 64 <- 32 channels)."""
    if dims == (128, 128, 128) and dt == torch.float16:
        pytest.skip("the full-size case runs once (bf16, the benchmarked type)")
    g = torch.Generator().manual_seed(cin + cout + sum(dims))
    d, h, w = dims
    w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    g3 = torch.randn(n, cout, d, h, w, generator=g)
    g1 = torch.randn(n, cout, d, h, w, generator=g)
    # d/dx of a stride-1 conv = conv_transpose of the gradient (no need to run a forward at 128^3 on the host)
    ref = F.conv_transpose3d(_rt(g3, dt), _rt(w3, dt), None, padding=1) + F.conv_transpose3d(_rt(g1, dt), _rt(w1, dt), None)
    p3 = ops.pack_weight(w3.to(DEV), N.ROLE_CONV_DGRAD, dt, 1)
    p1 = ops.pack_weight(w1.to(DEV), N.ROLE_CONV_DGRAD, dt, 1)
    gx = ops.conv_s1_dgrad_pair(ops.as_input(g3.to(DEV), dt), p3, ops.as_input(g1.to(DEV), dt), p1, (n, cin, d, h, w))
    assert gx is not None, "no fused kernel for %s" % (dims,)
    _close(gx, ref, EPS[dt], 1e-3, "s1 pair dgrad %s %s" % ((cin, cout), dims))


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,n,dims", [(64, 32, 2, (32, 32, 64)), (128, 64, 2, (32, 32, 32)), (64, 32, 2, (16, 40, 104)),
                                             (64, 32, 3, (16, 24, 64)), (128, 64, 2, (20, 40, 48)),
                                             (64, 32, 2, (128, 128, 128))])
def test_skip1x1_in_lrelu_fwd(dt, cin, cout, n, dims):
    """Decoder ResBlock tail (network.py:416 with skip = skip_conv(x)): lrelu(IN(y2) + conv1x1(x) + bias) in one pass.
    The skip conv's output is rounded to the storage type before the sum, like the stored tensor it replaces.  The last
    case is the benchmark's own shape (2 x 128^3, 64 -> 32)."""
    if dims == (128, 128, 128) and dt == torch.float16:
        pytest.skip("the full-size case runs once (bf16, the benchmarked type)")
    g = torch.Generator().manual_seed(cin + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    yv = torch.randn(n, cout, d, h, w, generator=g) * 2 + 0.3
    w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    b1 = torch.randn(cout, generator=g)
    mean = torch.randn(n * cout, generator=g) * 0.1
    scale = torch.rand(n * cout, generator=g) + 0.5
    out = ops.skip1x1_in_lrelu_fwd(ops.as_input(xv.to(DEV), dt), ops.pack_weight(w1.to(DEV), N.ROLE_CONV_FWD, dt, 1),
                                   b1.to(DEV), ops.as_input(yv.to(DEV), dt), mean.to(DEV), scale.to(DEV))
    assert out is not None, "no fused kernel for %s" % (dims,)
    skip = _rt(F.conv3d(_rt(xv, dt), _rt(w1, dt), b1), dt)
    ref = F.leaky_relu((_rt(yv, dt) - mean.view(n, cout, 1, 1, 1)) * scale.view(n, cout, 1, 1, 1) + skip, 0.01)
    # one output rounding + the rounding of the (never stored) skip value inside the sum
    _close(out, ref, 1.5 * EPS[dt], 1e-3, "fused skip tail %s %s" % ((cin, cout), dims))


def test_conv3_s1_32to32_full_size_random_data():
    """The benchmark's roofline kernel on the benchmark's own shape with random data: forward
    (+ bias + residual), input gradient and weight gradient of the 32 -> 32 3x3x3 conv on 2 x 128^3 against torch CPU on
    bf16-rounded operands.  (The ragged / small cases of the same kernels live in test_mfma_conv_s1_bf16.)"""
    n, c, s = 2, 32, 128
    g = torch.Generator().manual_seed(4128)
    xv = torch.randn(n, c, s, s, s, generator=g)
    wt = torch.randn(c, c, 3, 3, 3, generator=g) * (1.0 / (27 * c) ** 0.5)
    b = torch.randn(c, generator=g)
    r = torch.randn(n, c, s, s, s, generator=g)
    dt = torch.bfloat16
    x = ops.as_input(xv.to(DEV), dt)
    res = ops.as_input(r.to(DEV), dt)
    y = ops.conv_fwd(x, ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1), b.to(DEV), c, 3, 1, res=res)
    xr, wr = _rt(xv, dt), _rt(wt, dt)
    ref = F.conv3d(xr, wr, b, padding=1) + _rt(r, dt)
    _close(y, ref, 1.5 * EPS[dt], 1e-3, "32->32 fwd 2x128^3")
    del y, ref, res, r
    gy = torch.randn(n, c, s, s, s, generator=g)
    gyd = ops.as_input(gy.to(DEV), dt)
    gx = ops.conv_dgrad(gyd, ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, dt, 1), (n, c, s, s, s), 3, 1)
    gyr = _rt(gy, dt)
    _close(gx, F.conv_transpose3d(gyr, wr, None, padding=1), EPS[dt], 1e-3, "32->32 dgrad 2x128^3")
    del gx
    gw = ops.conv_wgrad(x, gyd, 3, 1)
    ref_w = torch.nn.grad.conv3d_weight(xr, wr.shape, gyr, padding=1)
    # fp32 accumulation over 4.2 M positions: relative to the gradient's max, not to one rounding
    _close(gw, ref_w, 2e-3, 1e-3, "32->32 wgrad 2x128^3")


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,n,dims,split", [(64, 32, 2, (64, 64, 64), False), (64, 32, 2, (64, 64, 64), True),
                                                   (128, 64, 2, (32, 32, 64), False), (64, 32, 2, (32, 64, 40), False),
                                                   (64, 32, 1, (64, 64, 96), True), (32, 32, 3, (64, 64, 32), False)])
def test_wgrad_pair(dt, cin, cout, n, dims, split):
    """Decoder ResBlock backward (network.py:403-409 read backwards): the weight gradients of conv1 (3x3x3) and skip_conv
    (1x1x1) on the same input from one pass over it (ru3d_conv3d_wgrad_pair: the sliding kernel's free tap slot), x dense
    or as the two planes of the level-0 concat, W a multiple of 32 or not.  The 3x3x3 part is bit for bit what the plain
    entry point returns (same kernel, same order); the 1x1x1 part is held against torch-CPU on storage-rounded operands."""
    g = torch.Generator().manual_seed(cin + cout + n + sum(dims) + int(split))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    g1 = torch.randn(n, cout, d, h, w, generator=g)
    g2 = torch.randn(n, cout, d, h, w, generator=g)
    if split:       # [u | skip] as two planes of one buffer
        buf = ops.as_input(torch.cat((xv[:, :cin // 2], xv[:, cin // 2:]), dim=0).to(DEV), dt)
        x = N.Split(buf)
    else:
        x = ops.as_input(xv.to(DEV), dt)
    dy1 = ops.as_input(g1.to(DEV), dt)
    dy2 = ops.as_input(g2.to(DEV), dt)
    both = ops.conv_wgrad_pair(x, dy1, dy2, 1)
    assert both is not None, "no fused kernel for %s" % (dims,)
    gw3, gw1 = both
    plain3 = ops.conv_wgrad(x, dy1, 3, 1)
    assert torch.equal(gw3, plain3), "3x3x3 part differs from the plain sliding kernel"
    ref1 = torch.nn.grad.conv3d_weight(_rt(xv, dt), (cout, cin, 1, 1, 1), _rt(g2, dt))
    # fp32 accumulation over up to 200 k positions per sample: relative to the gradient's maximum
    _close(gw1, ref1, 2e-3, 1e-3, "1x1x1 weight gradient from the pair kernel %s %s" % ((cin, cout), dims))
    if not split:
        plain1 = ops.conv_wgrad(x, dy2, 1, 1)
        _close(gw1, plain1, 1e-4, 1e-4, "pair vs staged 1x1x1 weight gradient")


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,n,dims", [(32, 64, 2, (64, 64, 64)), (64, 128, 2, (32, 64, 64)), (32, 64, 3, (32, 32, 96)),
                                             (128, 256, 2, (32, 32, 32)), (32, 32, 2, (64, 32, 64))])
def test_wgrad_pair_stride2(dt, cin, cout, n, dims):
    """Pooling ResBlock backward: the weight gradients of conv1 (3x3x3 stride 2) and skip_conv (1x1x1 stride 2) on the same
    input from one pass (the LDS-DMA stride-2 kernel's free tap slot: its centre tap gathers x[2 o], the rows the 1x1x1
    conv reads).  3x3x3 part bit for bit the plain kernel's, 1x1x1 part against torch-CPU."""
    g = torch.Generator().manual_seed(cin + cout + n + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    g1 = torch.randn(n, cout, d // 2, h // 2, w // 2, generator=g)
    g2 = torch.randn(n, cout, d // 2, h // 2, w // 2, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    dy1 = ops.as_input(g1.to(DEV), dt)
    dy2 = ops.as_input(g2.to(DEV), dt)
    both = ops.conv_wgrad_pair(x, dy1, dy2, 2)
    assert both is not None, "no fused kernel for %s" % (dims,)
    gw3, gw1 = both
    assert torch.equal(gw3, ops.conv_wgrad(x, dy1, 3, 2)), "3x3x3 part differs from the plain stride-2 kernel"
    ref1 = torch.nn.grad.conv3d_weight(_rt(xv, dt), (cout, cin, 1, 1, 1), _rt(g2, dt), stride=2)
    _close(gw1, ref1, 2e-3, 1e-3, "1x1x1 stride-2 weight gradient from the pair kernel %s %s" % ((cin, cout), dims))
    _close(gw1, ops.conv_wgrad(x, dy2, 1, 2), 1e-4, 1e-4, "pair vs staged 1x1x1 stride-2 weight gradient")
