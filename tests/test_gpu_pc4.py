"""Two kernels with 4 x 2 accumulator tiles per wave.  (1) The 512-voxel producer/consumer form of the 3x3x3 stride-1 conv (csrc/conv_mfma.hip conv3_s1_pc4_kernel; reference
network.py:394-416 at the 128-channel level of config 2 and the decoder's 128 -> 64 conv one level up): forward with bias
and residual, fused InstanceNorm statistics, input gradient (flipped taps) against torch CPU on operands rounded to the
storage type.  Every shape below satisfies the kernel's plan (D, H multiples of 4, width class 32, >= 192 workgroup
units), so the default routing runs it (RU3D_CONV_PC4=0 sends them back to the 256-voxel kernel, which the ragged cases
of tests/test_gpu_parity.py keep covering).
(2) conv3_s1_sk_kernel, the deep levels' form (16^3 x 256, 8^3 x 512 channels): one 128-voxel x 64-cout block per workgroup,
the input channels split over its four waves and, at 8^3, over workgroups too (fp32 partial slices + the fixed-order sum).
Run with `-m gpu`."""
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

# (n, cin, cout, dims): config 2's 128-channel level; ragged W (second tile 24 wide); three samples with several tiles per
# workgroup (384 tiles on 256 workgroups, sample changes inside a workgroup's run); the decoder's 256 -> 128 first conv;
# 64 couts with a 3-tile-wide row
SHAPES = [(2, 128, 128, (32, 32, 32)), (2, 64, 128, (16, 32, 56)), (3, 32, 64, (32, 32, 64)), (2, 256, 128, (32, 32, 32)),
          (1, 32, 64, (32, 48, 96))]


# sk: config 2's two deep levels, the decoder's 512 -> 256 conv at 16^3, a 32-wide case on the 16-wide tile with a split
# over workgroups, three samples with one 16-channel chunk per wave (ks = 4) on the 8-wide tile
# ragged extents (masked tiles): config 4's 20 x 20 x 10 and 10 x 10 x 5 levels, an odd D with a 13-wide row
SK_SHAPES = [(2, 256, 256, (16, 16, 16)), (2, 512, 512, (8, 8, 8)), (2, 512, 256, (16, 16, 16)), (2, 256, 256, (8, 8, 32)),
             (3, 256, 128, (4, 8, 24)), (2, 256, 256, (20, 20, 10)), (2, 512, 512, (10, 10, 5)), (2, 128, 192, (7, 9, 13))]


def _rt(t, dt):
    return t.to(dt).float()


def _close(a, b, rtol, atol, what):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * max(b.abs().max().item(), 1e-30)
    assert err <= lim, "%s: max err %.3e > %.3e" % (what, err, lim)


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,dims", SHAPES + SK_SHAPES)
def test_pc4_forward_bias_residual_and_dgrad(dt, n, cin, cout, dims):
    if dt == torch.float16 and cin * cout > 128 * 128 and dims[2] == 32 and dims[0] == 32:
        pytest.skip("the largest case runs once")
    g = torch.Generator().manual_seed(n + cin + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    r = torch.randn(n, cout, d, h, w, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    res = ops.as_input(r.to(DEV), dt)
    y = ops.conv_fwd(x, ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1), b.to(DEV), cout, 3, 1, res=res)
    xr, wr = _rt(xv, dt), _rt(wt, dt)
    plain = F.conv3d(xr, wr, b, padding=1)
    # two roundings: the conv's value, then the sum with the residual
    _close(y, _rt(plain, dt) + _rt(r, dt), 1.5 * EPS[dt], 1e-3, "pc4 fwd %s %s" % ((cin, cout), dims))
    y0 = ops.conv_fwd(x, ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1), None, cout, 3, 1)
    _close(y0, F.conv3d(xr, wr, None, padding=1), EPS[dt], 1e-3, "pc4 fwd, no bias %s %s" % ((cin, cout), dims))
    del y, y0, res
    gy = torch.randn(n, cout, d, h, w, generator=g)
    gx = ops.conv_dgrad(ops.as_input(gy.to(DEV), dt), ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, dt, 1),
                        (n, cin, d, h, w), 3, 1)
    # the input gradient has Cout = cin: it runs on this kernel when cin is a multiple of 64
    _close(gx, F.conv_transpose3d(_rt(gy, dt), wr, None, padding=1), EPS[dt], 1e-3, "pc4 dgrad %s %s" % ((cin, cout), dims))


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cout,dims", [SHAPES[0], SHAPES[1], SHAPES[2]])
def test_pc4_fused_instance_norm_statistics(dt, n, cin, cout, dims):
    """mean and 1/sqrt(var + eps) of the STORED conv output (network.py:384), summed in the kernel's epilogue and kept in
    LDS between a workgroup's tiles: against float64 statistics of the tensor the kernel wrote."""
    g = torch.Generator().manual_seed(7 * n + cin + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    y, mean, scale = ops.conv_fwd_in(x, ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1), b.to(DEV), cout, 3, 1)
    _close(y, F.conv3d(_rt(xv, dt), _rt(wt, dt), b, padding=1), EPS[dt], 1e-3, "pc4 fwd with statistics")
    yd = y.double()
    m = yd.mean(dim=(2, 3, 4)).reshape(-1)
    v = yd.var(dim=(2, 3, 4), unbiased=False).reshape(-1)
    _close(mean, m, 1e-5, 2e-5, "mean")
    _close(scale, 1.0 / (v + 1e-5).sqrt(), 2e-5, 1e-6, "scale")


def test_pc4_with_a_reduced_cu_budget():
    """N > 1 leaves CUs to RCCL (ru3d_set_cu_budget): the persistent grid shrinks, several tiles per workgroup, the
    statistics slab follows the grid - same result as with the whole chip."""
    n, cin, cout, dims = SHAPES[0]
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(99)
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, dt, 1)
    y0, m0, s0 = ops.conv_fwd_in(x, pw, b.to(DEV), cout, 3, 1)
    assert N.lib.ru3d_set_cu_budget(208) == 0
    try:
        y1, m1, s1 = ops.conv_fwd_in(x, pw, b.to(DEV), cout, 3, 1)
        torch.cuda.synchronize()
    finally:
        assert N.lib.ru3d_set_cu_budget(0) == 0
    assert torch.equal(y0, y1)
    _close(m1, m0, 1e-6, 1e-6, "mean under a reduced budget")
    _close(s1, s0, 1e-6, 1e-6, "scale under a reduced budget")
