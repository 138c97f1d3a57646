"""Time of the per-step weight packing of BASELINE config 2 (GPU box): python tools/packbench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")
ws = {c: (torch.randn(c, c, 3, 3, 3, device=dev), torch.randn(c, c, 3, 3, 3, device=dev)) for c in (32, 64, 128, 256, 512)}
def run():
    for c, reps in ((32, 3), (64, 4), (128, 5), (256, 6), (512, 5)):
        w1, w2 = ws[c]
        for _ in range(reps):
            ops.pack_weights([(w1, N.ROLE_CONV_FWD, 1), (w2, N.ROLE_CONV_FWD, 1), (w2, N.ROLE_CONV_DGRAD, 1), (w1, N.ROLE_CONV_DGRAD, 1)], torch.bfloat16)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("pack of a config-2-like weight set: %.3f ms" % (e0.elapsed_time(e1) / 10))
