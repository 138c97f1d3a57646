"""Diagnostic build only (make -C .../csrc stamps -> libru3d_stamps.so): cycle stamps of wave 0 of workgroup 0 of the
sliding 32->32 conv kernel at every (kd, kw, k-step) group boundary of four steady-state steps (one per ring phase)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RU3D_LIB"] = os.environ.get("RU3D_STAMPS_LIB") or os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd", "libru3d_stamps.so")
import torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
raw = ctypes.CDLL(os.environ["RU3D_LIB"])
dev = torch.device("cuda:0")
n, c, s = 2, 32, 128
x = torch.randn(n, s, s, s, c, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
b = torch.randn(c, device=dev)
pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
buf = torch.zeros(96, dtype=torch.int64, device=dev)
raw.ru3d_debug_slide_stamps.argtypes = [ctypes.c_void_p]
raw.ru3d_debug_slide_stamps(ctypes.c_void_p(buf.data_ptr()))
for mode in ("plain", "stats"):
    for _ in range(3):
        if mode == "plain": ops.conv_fwd(x, pw, b, c, 3, 1)
        else: ops.conv_fwd_in(x, pw, b, c, 3, 1)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    print(mode)
    for ph in range(4):
        st = t[ph * 24: ph * 24 + 19]; ex = t[ph * 24 + 19: ph * 24 + 23]
        d = [st[i + 1] - st[i] for i in range(18)]
        print("  phase %d: step %6d cyc | groups: %s" % (ph, st[18] - st[0], " ".join("%4d" % v for v in d)))
        print("           barrier group: mfma+epi %d | barrier %d | store_plane %d | load_plane %d | rest %d" % (
            ex[0] - st[11], ex[1] - ex[0], ex[2] - ex[1], ex[3] - ex[2], st[12] - ex[3]))
    nxt = [t[((ph + 1) % 4) * 24] - t[ph * 24 + 18] for ph in range(3)]
    print("  between steps (last stamp -> first stamp of next phase):", nxt)
