"""Micro-benchmark of the deep-level 3x3x3 stride-1 convs (bf16): python tools/kbench_deep.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
flt = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = [(128, (32, 32, 32)), (256, (16, 16, 16)), (512, (8, 8, 8)), (128, (40, 40, 20)), (256, (20, 20, 10)), (512, (10, 10, 5))]
for c, (d, h, w) in shapes:
    if flt and flt != str(c):
        continue
    x = torch.randn(2, d, h, w, c, device=dev).to(BF).permute(0, 4, 1, 2, 3)
    wt = torch.randn(c, c, 3, 3, 3, device=dev) * 0.02
    b = torch.randn(c, device=dev)
    pf = ops.pack_weight(wt, N.ROLE_CONV_FWD, BF, 1); pd = ops.pack_weight(wt, N.ROLE_CONV_DGRAD, BF, 1)
    fl = 2.0 * 2 * d * h * w * c * c * 27
    t = timeit(lambda: ops.conv_fwd(x, pf, b, c, 3, 1))
    t2 = timeit(lambda: ops.conv_dgrad(x, pd, tuple(x.shape), 3, 1))
    t3 = timeit(lambda: ops.conv_wgrad(x, x, 3, 1))
    print("%4d ch %2dx%2dx%2d  fwd %7.1f us (%4.1f%% of the MFMA peak)  dgrad %7.1f us  wgrad %7.1f us" % (c, d, h, w, t, fl / (t * 1e-6) / 2.5e15 * 100, t2, t3), flush=True)
