"""Throughput of the on-device patch sampling / augmentation (GPU box): python tools/augment_bench.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import augment
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
cases = [augment.DeviceCase(rng.randn(256, 256, 160, 1).astype(np.float32), (rng.rand(256, 256, 160) * 4).astype(np.uint8), dev)
         for _ in range(2)]
aug = augment.DeviceAugment(scale=0.1, crop_size=128, crop_mode="random")
for _ in range(3):
    aug.batch(cases, 2)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 50
for _ in range(n):
    aug.batch(cases, 2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("batch of 2 x 128^3 patches (rescale-crop + mirror + contrast + brightness + gamma): %.3f ms = %.0f M voxels/s"
      % (1e3 * dt, 2 * 128 ** 3 / dt / 1e6))
