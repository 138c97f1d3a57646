"""Diagnostic build only (make -C .../csrc stamps [STAMP_FLAGS=-DRU3D_SLIDE_STAMP_PAIRS STAMP_SUFFIX=_rows]): cycle stamps
of wave 0 of workgroup 0 of the sliding 32->32 conv kernel at every (kd, kw) pass - or every MFMA pair - of four
steady-state steps (one per ring phase).  A stamp is an s_memtime: it drains the wave's LDS queue, so the row-level
build perturbs more than the pass-level one; read both."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = len(sys.argv) > 1 and sys.argv[1] == "rows"
os.environ["RU3D_LIB"] = os.environ.get("RU3D_STAMPS_LIB") or os.path.join(
    ROOT, "3d-unet-renal-anatomy-extraction_amd", "libru3d_stamps_rows.so" if rows else "libru3d_stamps.so")
import torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
raw = ctypes.CDLL(os.environ["RU3D_LIB"])
dev = torch.device("cuda:0")
n, c, s = 2, 32, 128
x = torch.randn(n, s, s, s, c, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
r = torch.randn(n, s, s, s, c, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
b = torch.randn(c, device=dev)
pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
raw.ru3d_debug_slide_stamps.argtypes = [ctypes.c_void_p]
raw.ru3d_debug_slide_stamps(ctypes.c_void_p(buf.data_ptr()))
for mode in ("plain", "stats", "res"):
    for _ in range(3):
        if mode == "plain": ops.conv_fwd(x, pw, b, c, 3, 1)
        elif mode == "stats": ops.conv_fwd_in(x, pw, b, c, 3, 1)
        else: ops.conv_fwd(x, pw, b, c, 3, 1, res=r)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    print(mode)
    for ph in range(4):
        st = t[ph * 128: ph * 128 + 128]
        if rows:
            pts = st[:108] + [st[127]]
            d = [pts[i + 1] - pts[i] for i in range(108)]
            print("  phase %d: step %6d cyc" % (ph, st[127] - st[0]))
            for q in range(9):
                print("     pass %d: %5d | pairs %s" % (q, sum(d[q * 12:q * 12 + 12]), " ".join("%3d" % v for v in d[q * 12:q * 12 + 12])))
        else:
            pts = st[:9] + [st[127]]
            d = [pts[i + 1] - pts[i] for i in range(9)]
            print("  phase %d: step %6d cyc | passes: %s" % (ph, st[127] - st[0], " ".join("%4d" % v for v in d)))
    # s_memrealtime ticks at 100 MHz: core clock over phases 0..2 = shader cycles / real time
    cyc = t[2 * 128 + 127] - t[0]
    rt = (t[2 * 128 + 121] - t[120]) / 100e6
    if rt > 0:
        print("  core clock over these steps: %.2f GHz" % (cyc / rt / 1e9))
    nxt = [t[((ph + 1) % 4) * 128] - t[ph * 128 + 127] for ph in range(3)]
    print("  between steps (last stamp -> first stamp of next phase):", nxt)
