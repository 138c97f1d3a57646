"""Micro-benchmark of individual conv kernels on the GPU box (bf16): python tools/kbench.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

cases = [(2, 32, 32, 128), (2, 64, 32, 128), (2, 64, 64, 64), (2, 128, 64, 64), (2, 128, 128, 32), (2, 256, 256, 16), (2, 512, 512, 8)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for n, cin, cout, s in cases:
    x = torch.randn(n, s, s, s, cin, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    dy = torch.randn(n, s, s, s, cout, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    flops = 2.0 * n * s ** 3 * 27 * cin * cout
    byts = (n * s ** 3 * (cin + cout) + 27 * cin * cout) * 2
    t = timeit(lambda: ops.conv_fwd(x, pw, b, cout, 3, 1))
    tw = timeit(lambda: ops.conv_wgrad(x, dy, 3, 1))
    print("conv3 s1 %3d->%3d @%3d^3: fwd %.3f ms  %6.0f TF/s (%.1f%% mfma) %5.0f GB/s alg (%.1f%% hbm) | wgrad %.3f ms %6.0f TF/s" %
          (cin, cout, s, t, flops / t / 1e9, flops / t / 1e9 / 2500 * 100, byts / t / 1e6, byts / t / 1e6 / 8000 * 100, tw, flops / tw / 1e9), flush=True)
