"""Throughput of trainer.predict_per_patch on the GPU box: config-2 model, bf16, one 256x256x160 case, 128^3 windows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import numpy as np, torch, network, trainer as T, inference as I
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = network.ResUnet3D(4, 32, 1, 3).to(dev)
network.set_compute_dtype(model, torch.bfloat16)
vol = np.random.default_rng(0).standard_normal((256, 256, 160, 1)).astype(np.float32)
for spp, pb in ((2, 1), (2, 2), (4, 2)):
    origins, counts = I.window_origins(I.padded_shape(vol.shape[:3], (128,) * 3), (128,) * 3, spp)
    T.predict_per_patch(vol, model, 3, (128, 128, 128), spp, False, False, patch_batch=pb)   # warm-up (packs cached after)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mask = T.predict_per_patch(vol, model, 3, (128, 128, 128), spp, False, False, patch_batch=pb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("step_per_patch %d patch_batch %d: %d windows %s, %.1f ms total, %.2f ms/window, %.1f M window-voxels/s" % (
        spp, pb, len(origins), counts, 1e3 * dt, 1e3 * dt / len(origins), len(origins) * 128 ** 3 / dt / 1e6))
