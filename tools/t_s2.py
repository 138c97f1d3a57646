"""Parity of the stride-2 tile kernels against torch CPU on bf16-rounded operands (quick dev check; the -m gpu suite
holds the real tests)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
DEV = torch.device("cuda:0")
BF = torch.bfloat16
F = torch.nn.functional


def close(got, ref, rel, name):
    got = got.float().cpu()
    err = (got - ref).abs().max().item(); m = ref.abs().max().item()
    good = err <= rel * m + 1e-3
    print("%-50s err %.4g  max %.3g  %s" % (name, err, m, "ok" if good else "FAIL"), flush=True)
    return good


ok = True
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "t"):
    for cin, cout, dims in [(64, 32, (16, 15, 33)), (64, 32, (16, 16, 32)), (64, 32, (9, 20, 40)), (64, 64, (8, 16, 48)),
                            (64, 32, (32, 32, 32))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (27 * cin / 8) ** 0.5)
        b = torch.randn(cout, generator=g)
        x = ops.as_input(xv.to(DEV), BF)
        pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_FWD, BF)
        y = ops.convt_fwd(x, pw, b.to(DEV), cout)
        xr = xv.bfloat16().float(); wr = wt.bfloat16().float()
        ref = F.pad(F.conv_transpose3d(xr, wr, b, stride=2, padding=1), (0, 1, 0, 1, 0, 1))
        ok &= close(y, ref, 2 ** -8, "convT fwd %s %s" % ((cin, cout), dims))
        yc = y.float().cpu()
        if not (float(yc[:, :, -1].abs().max()) == 0 and float(yc[:, :, :, -1].abs().max()) == 0 and float(yc[..., -1].abs().max()) == 0):
            print("far planes not zero"); ok = False
    # stride-2 conv input gradient with residual, odd extents
    for cin, cout, dims in [(32, 64, (31, 32, 66)), (32, 64, (32, 32, 64)), (32, 64, (63, 64, 98)), (64, 64, (16, 24, 40))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
        xr = xv.bfloat16().float().requires_grad_(True); wr = wt.bfloat16().float()
        ref = F.conv3d(xr, wr, None, stride=2, padding=1)
        gy = torch.randn(ref.shape, generator=g); rv = torch.randn(xv.shape, generator=g)
        gyd = ops.as_input(gy.to(DEV), BF); res = ops.as_input(rv.to(DEV), BF)
        pwd = ops.pack_weight(wt.to(DEV), N.ROLE_CONV_DGRAD, BF, 2)
        gx = ops.conv_dgrad(gyd, pwd, tuple(xv.shape), 3, 2, res=res)
        ref.backward(gy.bfloat16().float())
        ok &= close(gx, xr.grad + rv.bfloat16().float(), 2 ** -8, "s2 dgrad+res %s %s" % ((cin, cout), dims))
if which in ("all", "p"):
    # pooling block: both stride-2 input gradients + residual in one launch
    for cin, cout, dims in [(32, 64, (32, 32, 64)), (32, 64, (31, 30, 66)), (32, 64, (64, 64, 64))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
        w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        xr = xv.bfloat16().float().requires_grad_(True)
        y3 = F.conv3d(xr, w3.bfloat16().float(), None, stride=2, padding=1)
        y1 = F.conv3d(xr, w1.bfloat16().float(), None, stride=2)
        g3 = torch.randn(y3.shape, generator=g); g1 = torch.randn(y1.shape, generator=g); rv = torch.randn(xv.shape, generator=g)
        (y3 * g3.bfloat16().float()).sum().backward(retain_graph=True)
        (y1 * g1.bfloat16().float()).sum().backward()
        ref = xr.grad + rv.bfloat16().float()
        p3 = ops.pack_weight(w3.to(DEV), N.ROLE_CONV_DGRAD, BF, 2); p1 = ops.pack_weight(w1.to(DEV), N.ROLE_CONV_DGRAD, BF, 2)
        gx = ops.conv_s2_dgrad_pair(ops.as_input(g3.to(DEV), BF), p3, ops.as_input(g1.to(DEV), BF), p1, tuple(xv.shape),
                                    res=ops.as_input(rv.to(DEV), BF))
        if gx is None:
            print("pair dgrad %s: no fused kernel" % (dims,)); ok = False; continue
        ok &= close(gx, ref, 2 ** -8, "s2 pair dgrad+res %s %s" % ((cin, cout), dims))
        gx = ops.conv_s2_dgrad_pair(ops.as_input(g3.to(DEV), BF), p3, ops.as_input(g1.to(DEV), BF), p1, tuple(xv.shape))
        ok &= close(gx, xr.grad, 2 ** -8, "s2 pair dgrad %s %s" % ((cin, cout), dims))
    # ConvTranspose + fused InstanceNorm statistics
    for cin, cout, dims in [(64, 32, (16, 15, 33)), (64, 32, (32, 32, 32)), (64, 64, (8, 16, 48))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(3, cin, d, h, w, generator=g)
        wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (27 * cin / 8) ** 0.5)
        b = torch.randn(cout, generator=g)
        y, mean, scale = ops.convt_fwd_in(ops.as_input(xv.to(DEV), BF), ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_FWD, BF), b.to(DEV), cout)
        yf = y.float().cpu().double()
        m = yf.mean(dim=(2, 3, 4)).reshape(-1); v = yf.var(dim=(2, 3, 4), unbiased=False).reshape(-1)
        e1 = (mean.cpu().double() - m).abs().max().item(); e2 = (scale.cpu().double() - 1.0 / (v + 1e-5).sqrt()).abs().max().item()
        good = e1 < 1e-5 and e2 < 1e-4
        print("convT fwd_in stats %s: mean err %.3g scale err %.3g %s" % (dims, e1, e2, "ok" if good else "FAIL")); ok &= good
if which in ("all", "g"):
    # stride-2 conv forward (bias), odd extents and ragged tiles; ConvTranspose input gradient (the same form over dy)
    for cin, cout, dims in [(32, 64, (32, 32, 64)), (32, 64, (31, 30, 66)), (32, 64, (63, 64, 98)), (32, 128, (16, 24, 40)),
                            (32, 64, (64, 64, 64))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
        b = torch.randn(cout, generator=g)
        y = ops.conv_fwd(ops.as_input(xv.to(DEV), BF), ops.pack_weight(wt.to(DEV), N.ROLE_CONV_FWD, BF, 2), b.to(DEV), cout, 3, 2)
        ref = F.conv3d(xv.bfloat16().float(), wt.bfloat16().float(), b, stride=2, padding=1)
        ok &= close(y, ref, 2 ** -8, "s2 fwd %s %s" % ((cin, cout), dims))
    for cin, cout, dims in [(64, 32, (16, 16, 32)), (64, 32, (16, 15, 33)), (64, 32, (32, 32, 32))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (27 * cin / 8) ** 0.5)
        xr = xv.bfloat16().float().requires_grad_(True)
        ref = F.pad(F.conv_transpose3d(xr, wt.bfloat16().float(), None, stride=2, padding=1), (0, 1, 0, 1, 0, 1))
        gy = torch.randn(ref.shape, generator=g)
        gy[:, :, -1] = 0; gy[:, :, :, -1] = 0; gy[..., -1] = 0
        gx = ops.convt_dgrad(ops.as_input(gy.to(DEV), BF), ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_DGRAD, BF), tuple(xv.shape))
        ref.backward(gy.bfloat16().float())
        ok &= close(gx, xr.grad, 2 ** -8, "convT dgrad %s %s" % ((cin, cout), dims))
if which in ("all", "f"):
    # pooling block forward: conv1 + statistics (with a Dropout3d factor) + skip conv in one launch
    for cin, cout, dims in [(32, 64, (32, 32, 64)), (32, 64, (31, 30, 66)), (32, 64, (64, 64, 64))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(3, cin, d, h, w, generator=g)
        w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
        w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        b3 = torch.randn(cout, generator=g); b1 = torch.randn(cout, generator=g)
        drop = (torch.rand(3 * cout, generator=g) > 0.5).float() * 2.0
        x = ops.as_input(xv.to(DEV), BF)
        out = ops.conv_s2_pair_fwd_in(x, ops.pack_weight(w3.to(DEV), N.ROLE_CONV_FWD, BF, 2), b3.to(DEV),
                                      ops.pack_weight(w1.to(DEV), N.ROLE_CONV_FWD, BF, 2), b1.to(DEV), cout, drop.to(DEV))
        if out is None:
            print("pair fwd %s: no fused kernel" % (dims,)); ok = False; continue
        y3, mean, scale, y1 = out
        r3 = F.conv3d(xv.bfloat16().float(), w3.bfloat16().float(), b3, stride=2, padding=1)
        r1 = F.conv3d(xv.bfloat16().float(), w1.bfloat16().float(), b1, stride=2)
        ok &= close(y3, r3, 2 ** -8, "pair fwd conv3 %s" % (dims,))
        ok &= close(y1, r1, 2 ** -8, "pair fwd conv1 %s" % (dims,))
        yf = y3.float().cpu().double()
        m = yf.mean(dim=(2, 3, 4)).reshape(-1); v = yf.var(dim=(2, 3, 4), unbiased=False).reshape(-1)
        sd = drop.double()
        e1 = (mean.cpu().double() - m).abs().max().item(); e2 = (scale.cpu().double() - sd / (sd * sd * v + 1e-5).sqrt()).abs().max().item()
        good = e1 < 1e-5 and e2 < 1e-4
        print("pair fwd stats %s: mean err %.3g scale err %.3g %s" % (dims, e1, e2, "ok" if good else "FAIL")); ok &= good
if which in ("all", "k"):
    # decoder tail: lrelu(IN(y2) + conv1x1(x)) in one pass
    for cin, cout, dims in [(64, 32, (32, 32, 64)), (128, 64, (32, 32, 32)), (64, 32, (16, 40, 104))]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(2, cin, d, h, w, generator=g)
        yv = torch.randn(2, cout, d, h, w, generator=g) * 2 + 0.3
        w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        b1 = torch.randn(cout, generator=g)
        mean = torch.randn(2 * cout, generator=g) * 0.1; scale = torch.rand(2 * cout, generator=g) + 0.5
        out = ops.skip1x1_in_lrelu_fwd(ops.as_input(xv.to(DEV), BF), ops.pack_weight(w1.to(DEV), N.ROLE_CONV_FWD, BF, 1), b1.to(DEV),
                                       ops.as_input(yv.to(DEV), BF), mean.to(DEV), scale.to(DEV))
        if out is None:
            print("fused skip %s: no kernel" % (dims,)); ok = False; continue
        skip = F.conv3d(xv.bfloat16().float(), w1.bfloat16().float(), b1).bfloat16().float()
        ref = F.leaky_relu((yv.bfloat16().float() - mean.view(2, cout, 1, 1, 1)) * scale.view(2, cout, 1, 1, 1) + skip, 0.01)
        ok &= close(out, ref, 2 ** -8, "fused skip tail %s %s" % ((cin, cout), dims))
if which in ("all", "d"):
    # decoder block: conv1 (k3 s1) + skip conv (k1 s1) input gradients in one launch (32 -> 64 / 32 channels)
    for cin, cout, dims, n in [(64, 32, (16, 64, 64), 2), (32, 32, (8, 64, 128), 3), (64, 32, (32, 32, 64), 2), (64, 32, (16, 64, 80), 2)]:
        g = torch.Generator().manual_seed(cin + cout + sum(dims))
        d, h, w = dims
        xv = torch.randn(n, cin, d, h, w, generator=g)
        w3 = torch.randn(cout, cin, 3, 3, 3, generator=g) * (1.0 / (27 * cin) ** 0.5)
        w1 = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin ** 0.5)
        xr = xv.bfloat16().float().requires_grad_(True)
        y3 = F.conv3d(xr, w3.bfloat16().float(), None, padding=1)
        y1 = F.conv3d(xr, w1.bfloat16().float(), None)
        g3 = torch.randn(y3.shape, generator=g); g1 = torch.randn(y1.shape, generator=g)
        (y3 * g3.bfloat16().float()).sum().backward(retain_graph=True)
        (y1 * g1.bfloat16().float()).sum().backward()
        p3 = ops.pack_weight(w3.to(DEV), N.ROLE_CONV_DGRAD, BF, 1); p1 = ops.pack_weight(w1.to(DEV), N.ROLE_CONV_DGRAD, BF, 1)
        gx = ops.conv_s1_dgrad_pair(ops.as_input(g3.to(DEV), BF), p3, ops.as_input(g1.to(DEV), BF), p1, tuple(xv.shape))
        if gx is None:
            print("s1 pair dgrad %s: no fused kernel" % (dims,)); ok = False; continue
        ok &= close(gx, xr.grad, 2 ** -8, "s1 pair dgrad %s %s" % ((cin, cout), dims))
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
