"""Step cost of the sliding 32->32 kernel without stamps: time the launch at several depths D (the work per workgroup is
D/2 steps at 2x(D,128,128)); the slope is the steady-state cost per step, the intercept the prologue + tail."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")
c = 32
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
b = torch.randn(c, device=dev)
pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
res = {}
for D in (32, 64, 128, 256, 128, 64, 32):
    x = torch.randn(2, D, 128, 128, c, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    for _ in range(10): ops.conv_fwd(x, pw, b, c, 3, 1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): ops.conv_fwd(x, pw, b, c, 3, 1)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 40
    res.setdefault(D, []).append(t)
    print("D=%3d: %.4f ms  (%d steps per workgroup)  %.0f TF/s" % (D, t, D // 2, 2.0 * 2 * D * 128 * 128 * 27 * c * c / t / 1e9), flush=True)
a, b2 = min(res[256]), min(res[128])
print("slope: %.1f ns per step; intercept (prologue + tail + launch): %.1f us" % ((a - b2) / 64 * 1e6, (b2 - (a - b2)) * 1e3))
