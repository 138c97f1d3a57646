"""Micro-benchmark of the data-movement conv forms of config 2's level 0/1 (bf16, random data): python tools/kbench2.py
   stride-2 3x3x3 conv 32->64 (128^3 -> 64^3) forward / input gradient, ConvTranspose k3 s2 64->32 (64^3 -> 128^3) forward /
   input gradient, 1x1x1 convs 64->32 and 32->64 at 128^3."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")

def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def act(n, c, s):
    return torch.randn(n, s, s, s, c, device=dev).bfloat16().permute(0, 4, 1, 2, 3)

x128_32, x128_64, x64_64 = act(2, 32, 128), act(2, 64, 128), act(2, 64, 64)
b64, b32 = torch.randn(64, device=dev), torch.randn(32, device=dev)
w_dn = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05
w_up = torch.randn(64, 32, 3, 3, 3, device=dev) * 0.05          # ConvTranspose3d weight [cin, cout, k, k, k]
w_11a = torch.randn(32, 64, 1, 1, 1, device=dev) * 0.1
w_11b = torch.randn(64, 32, 1, 1, 1, device=dev) * 0.1
pw_dn = ops.pack_weight(w_dn, N.ROLE_CONV_FWD, torch.bfloat16, 2)
pw_dn_d = ops.pack_weight(w_dn, N.ROLE_CONV_DGRAD, torch.bfloat16, 2)
pw_up = ops.pack_weight(w_up, N.ROLE_CONVT_FWD, torch.bfloat16, 2)
pw_up_d = ops.pack_weight(w_up, N.ROLE_CONVT_DGRAD, torch.bfloat16, 2)
pw_11a = ops.pack_weight(w_11a, N.ROLE_CONV_FWD, torch.bfloat16, 1)
pw_11b = ops.pack_weight(w_11b, N.ROLE_CONV_FWD, torch.bfloat16, 1)
MB = 1e6
cases = [
    ("wgrad k3 s2 32->64 (128^3 x, 64^3 dy)", lambda: ops.conv_wgrad(x128_32, x64_64, 3, 2), (268.4 + 67.1)),
    ("convT wgrad 64->32 (64^3 x, 128^3 dy)", lambda: ops.convt_wgrad(x64_64, x128_32), (268.4 + 67.1)),
    ("conv k3 s2 32->64 fwd   128^3->64^3", lambda: ops.conv_fwd(x128_32, pw_dn, b64, 64, 3, 2), (268.4 + 67.1)),
    ("conv k3 s2 32->64 dgrad 64^3->128^3", lambda: ops.conv_dgrad(x64_64, pw_dn_d, (2, 32, 128, 128, 128), 3, 2), (67.1 + 268.4)),
    ("convT k3 s2 64->32 fwd  64^3->128^3", lambda: ops.convt_fwd(x64_64, pw_up, b32, 32), (67.1 + 268.4)),
    ("convT k3 s2 64->32 dgrad 128^3->64^3", lambda: ops.convt_dgrad(x128_32, pw_up_d, (2, 64, 64, 64, 64)), (268.4 + 67.1)),
    ("conv k1 64->32 @128^3", lambda: ops.conv_fwd(x128_64, pw_11a, b32, 32, 1, 1), (536.9 + 268.4)),
    ("conv k1 32->64 @128^3", lambda: ops.conv_fwd(x128_32, pw_11b, b64, 64, 1, 1), (268.4 + 536.9)),
]
for name, fn, mb in cases:
    t = timeit(fn)
    print("%-40s %.3f ms   %5.0f GB/s algorithmic" % (name, t, mb / t), flush=True)
