"""Micro-benchmark of the stride-2 / transposed / 1x1x1 conv forms at the four level boundaries of config 2 (bf16):
    python tools/kbench_direct.py [filter]
Prints microseconds per launch and the algorithmic bytes (|in| + |out| [+ |res|]) per second."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")
BF = torch.bfloat16
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def act(n, c, s):
    return torch.randn(n, s, s, s, c, device=dev).to(BF).permute(0, 4, 1, 2, 3)


def report(name, us, nbytes):
    print("%-46s %8.1f us  %6.0f GB/s alg  (%4.1f%% of 8 TB/s)" % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80), flush=True)


for lvl, (c, s) in enumerate([(32, 128), (64, 64), (128, 32), (256, 16)]):
    n = 2
    tag = "L%d%d " % (lvl, lvl + 1)
    xb = act(n, c, s)              # big side, C channels
    xs = act(n, 2 * c, s // 2)     # small side, 2C channels
    big_b, small_b = xb.numel() * 2, xs.numel() * 2
    w3 = torch.randn(2 * c, c, 3, 3, 3, device=dev) * 0.05
    w1 = torch.randn(2 * c, c, 1, 1, 1, device=dev) * 0.05
    wt = torch.randn(2 * c, c, 3, 3, 3, device=dev) * 0.05      # ConvTranspose [Cin = 2C, Cout = C]
    b2 = torch.randn(2 * c, device=dev); b1 = torch.randn(c, device=dev)
    p3f = ops.pack_weight(w3, N.ROLE_CONV_FWD, BF, 2); p3d = ops.pack_weight(w3, N.ROLE_CONV_DGRAD, BF, 2)
    p1f = ops.pack_weight(w1, N.ROLE_CONV_FWD, BF, 2); p1d = ops.pack_weight(w1, N.ROLE_CONV_DGRAD, BF, 2)
    ptf = ops.pack_weight(wt, N.ROLE_CONVT_FWD, BF); ptd = ops.pack_weight(wt, N.ROLE_CONVT_DGRAD, BF)
    shp = (n, c, s, s, s)
    cases = [
        ("conv3 s2 fwd %d->%d @%d^3" % (c, 2 * c, s), lambda: ops.conv_fwd(xb, p3f, b2, 2 * c, 3, 2), big_b + small_b),
        ("conv3 s2 dgrad %d->%d (+res)" % (2 * c, c), lambda: ops.conv_dgrad(xs, p3d, shp, 3, 2, res=xb), 2 * big_b + small_b),
        ("conv1 s2 fwd %d->%d" % (c, 2 * c), lambda: ops.conv_fwd(xb, p1f, b2, 2 * c, 1, 2), big_b // 8 + small_b),
        ("conv1 s2 dgrad %d->%d (+res)" % (2 * c, c), lambda: ops.conv_dgrad(xs, p1d, shp, 1, 2, res=xb), 2 * big_b + small_b),
        ("convT fwd %d->%d @%d^3" % (2 * c, c, s // 2), lambda: ops.convt_fwd(xs, ptf, b1, c), big_b + small_b),
        ("convT dgrad %d->%d" % (c, 2 * c), lambda: ops.convt_dgrad(xb, ptd, tuple(xs.shape)), big_b + small_b),
    ]
    # 1x1 stride-1 convs of the decoder block at this level: 2C -> C forward, C -> 2C input gradient
    xc = act(n, 2 * c, s)
    w11 = torch.randn(c, 2 * c, 1, 1, 1, device=dev) * 0.05
    q1f = ops.pack_weight(w11, N.ROLE_CONV_FWD, BF, 1); q1d = ops.pack_weight(w11, N.ROLE_CONV_DGRAD, BF, 1)
    cases += [
        ("conv1 s1 fwd %d->%d @%d^3" % (2 * c, c, s), lambda: ops.conv_fwd(xc, q1f, b1, c, 1, 1), 3 * big_b),
        ("conv1 s1 dgrad %d->%d" % (c, 2 * c), lambda: ops.conv_dgrad(xb, q1d, tuple(xc.shape), 1, 1), 3 * big_b),
    ]
    for name, fn, nb in cases:
        if flt and flt != "all" and flt not in (tag + name):
            continue
        report(tag + name, timeit(fn), nb)
