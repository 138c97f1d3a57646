"""Child of tests/test_gpu_switches.py: one training forward + backward of the config-2 architecture on a reduced patch
under whatever RU3D_* kernel switches the environment carries (they are read once, when libru3d.so first needs them):
    python switch_child.py <out.pt> [bn]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import loss as L  # noqa: E402
import network  # noqa: E402
import _ops as ops  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402


def main():
    out, kind = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "in")
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    if kind == "bn":
        model = network.ResAttrBNUnet3D(2, 32, 1, 3).to(dev)
        shape = (2, 1, 32, 32, 32)
    elif kind == "c4":
        model = network.ResUnet3D(4, 30, 1, 3).to(dev)
        shape = (2, 1, 80, 80, 48)          # halves to 40 / 20 / 10 / 5 x 3: no level fits a tile
    else:
        model = network.ResUnet3D(4, 32, 1, 3).to(dev)
        shape = (2, 1, 64, 64, 64)
    # (BatchNorm: fp32 storage on both sides - the comparison is native kernels against torch modules, not 16-bit rounding)
    network.set_compute_dtype(model, {"c4": torch.float16, "bn": torch.float32}.get(kind, torch.bfloat16))
    # BatchNorm: training mode (batch statistics); the others with Dropout3d off, so that the parent can hold every run
    # against the CPU oracle's gradients
    model.train(kind == "bn")
    if kind == "bn":        # the torch-module path draws its Dropout3d masks from torch's generator, the native one from its own
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout3d):
                m.p = 0.0
    ops._drop_counter[0] = 0
    x = O.synth_image(shape, 5).to(dev)
    y = O.phantom_labels(shape[0], shape[2:], 3).to(dev)
    logits = model(x)
    loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
    # fp16 storage needs the loss scale of the training loop (optim.LossScaler / the reference's apex O1): unscaled, the
    # activation gradients of a 1.2 M-voxel patch are fp16 subnormals
    scale = 65536.0 if kind == "c4" else 1.0
    (loss * scale).backward()
    torch.cuda.synchronize()
    torch.save({"loss": float(loss.detach()), "logits": logits.detach().float().cpu(),
                "grads": {k: (p.grad.detach().float() / scale).cpu() for k, p in model.named_parameters()
                          if p.grad is not None}}, out)


if __name__ == "__main__":
    main()
