"""The head's fused backward (ru3d_head_bwd: dx, dW, db from one pass over the head's input and the fp32 gradient of the
logits; reference network.py:547 `fc` under autograd) and the stem's weight gradient with the bias gradient as a 28th row
(ru3d_conv3d_wgrad_bias; network.py:541 `conv`), against torch CPU on the operands rounded to the storage type.
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


def _close(a, b, rtol, atol, what):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    lim = atol + rtol * max(b.abs().max().item(), 1e-30)
    assert err <= lim, "%s: max err %.3e > %.3e" % (what, err, lim)


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,cin,cin_real,cout,dims", [(2, 32, 32, 3, (16, 24, 40)), (1, 32, 30, 2, (9, 10, 11)),
                                                      (3, 64, 64, 4, (8, 8, 8)), (2, 32, 32, 1, (5, 7, 33)),
                                                      (2, 32, 32, 3, (64, 64, 64))])
def test_head_bwd_fused(dt, n, cin, cin_real, cout, dims):
    g = torch.Generator().manual_seed(n + cin + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, cin, d, h, w, generator=g)
    xv[:, cin_real:] = 0                                  # pad lanes of a channel-padded net hold exact zeros
    wt = torch.randn(cout, cin_real, 1, 1, 1, generator=g) * 0.3
    gv = torch.randn(n, cout, d, h, w, generator=g) * 1e-2
    x = ops.as_input(xv.to(DEV), dt)
    gy = N.to_ndhwc(gv.to(DEV))                           # fp32, channels last: how the loss kernel leaves dlogits
    out = ops.head_bwd(x, gy, wt.to(DEV), True)
    assert out is not None
    dx, dw, db = out
    xr = xv.to(dt).double()
    gr = gv.to(dt).double()                               # the rounding the unfused path's cast stored
    wr = torch.zeros(cout, cin, dtype=torch.float64)
    wr[:, :cin_real] = wt.to(dt).double().view(cout, cin_real)
    ref_dx = torch.einsum("ncdhw,ck->nkdhw", gr, wr)
    _close(dx, ref_dx, EPS[dt], 1e-6, "dx")
    assert float(dx[:, cin_real:].abs().max()) == 0.0 if cin_real < cin else True
    ref_dw = torch.einsum("nkdhw,ncdhw->ck", xr, gr)[:, :cin_real]
    _close(dw.view(cout, cin_real), ref_dw, 2e-5, 1e-6, "dW")
    _close(db, gr.sum(dim=(0, 2, 3, 4)), 2e-5, 1e-6, "db")


@pytest.mark.parametrize("dt", DTYPES, ids=["bf16", "fp16"])
@pytest.mark.parametrize("n,cout,dims", [(2, 32, (16, 24, 40)), (1, 32, (9, 10, 35)), (2, 64, (8, 16, 32)), (2, 32, (64, 64, 64))])
def test_stem_wgrad_with_bias_row(dt, n, cout, dims):
    g = torch.Generator().manual_seed(n + cout + sum(dims))
    d, h, w = dims
    xv = torch.randn(n, 1, d, h, w, generator=g)
    gv = torch.randn(n, cout, d, h, w, generator=g)
    x = ops.as_input(xv.to(DEV), dt)
    gy = ops.as_input(gv.to(DEV), dt)
    dw, db = ops.conv_wgrad_bias(x, gy, 3, 1)
    xr, gr = xv.to(dt).float(), gv.to(dt).float()
    ref_w = torch.nn.grad.conv3d_weight(xr, (cout, 1, 3, 3, 3), gr, padding=1)
    _close(dw, ref_w, 2e-3, 1e-3, "stem dW")
    _close(db, gr.double().sum(dim=(0, 2, 3, 4)), 1e-4, 1e-3, "stem db")
    assert torch.equal(dw, ops.conv_wgrad(x, gy, 3, 1))           # the bias row leaves the weight rows' bits alone
