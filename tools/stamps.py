"""Diagnostic build only (libru3d_stamps.so): where do the producer / consumer waves of conv3_s1_pc spend cycles?"""
import ctypes, os, sys
os.environ["RU3D_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-unet-renal-anatomy-extraction_amd", "libru3d_stamps.so")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
raw = ctypes.CDLL(os.environ["RU3D_LIB"])
dev = torch.device("cuda:0")
for n, cin, cout, s in [(2, 32, 32, 128), (2, 64, 64, 64), (2, 128, 64, 64)]:
    x = torch.randn(n, s, s, s, cin, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    for _ in range(3): ops.conv_fwd(x, pw, None, cout, 3, 1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    raw.ru3d_debug_stamps(buf, 1)
    R = 5
    for _ in range(R): ops.conv_fwd(x, pw, None, cout, 3, 1)
    torch.cuda.synchronize()
    raw.ru3d_debug_stamps(buf, 1)
    cw, iters = buf[4], buf[5]
    print("conv %d->%d @%d^3: consumer waves %d, iterations/wave %.1f" % (cin, cout, s, cw, iters / max(cw, 1)))
    tot = sum(buf[i] for i in range(4))
    for i, nm in enumerate(["weight prologue+zero", "mfma loop", "epilogue", "barrier wait"]):
        print("   consumer %-22s %9.0f cyc/iter %5.1f%%" % (nm, buf[i] / max(iters, 1), 100.0 * buf[i] / max(tot, 1)))
    pw_ = buf[10]
    ptot = buf[8] + buf[9]
    print("   producer load+write      %9.0f cyc/iter %5.1f%% | barrier wait %9.0f %5.1f%%" % (buf[8] / max(iters, 1), 100.0 * buf[8] / max(ptot, 1), buf[9] / max(iters, 1), 100.0 * buf[9] / max(ptot, 1)))
