import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")
x = ops.as_input(torch.randn(2, 1, 128, 128, 128, device=dev), torch.bfloat16)
w = torch.randn(32, 1, 3, 3, 3, device=dev) * 0.2
b = torch.randn(32, device=dev)
pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
for _ in range(5): y = ops.conv_fwd(x, pw, b, 32, 3, 1)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): y = ops.conv_fwd(x, pw, b, 32, 3, 1)
e1.record(); torch.cuda.synchronize()
ref = torch.nn.functional.conv3d(x.float(), w.bfloat16().float(), b, padding=1)
print("stem fwd %.4f ms  max err %.4f" % (e0.elapsed_time(e1) / 30, (y.float() - ref).abs().max().item()))
