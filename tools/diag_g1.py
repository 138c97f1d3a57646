"""Diagnostic (GPU box): error statistics of the native path vs the G1 golden in fp32 and bf16 mode."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import network, loss as L
z = np.load(os.path.join(ROOT, "tests/golden/g1_config1.npz"))
dev = torch.device("cuda:0")
for dt in (torch.float32, torch.bfloat16):
    m = network.ResUnet3D(2, 8, 1, 2)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")})
    m = m.to(dev).eval(); network.set_compute_dtype(m, dt)
    x = torch.from_numpy(z["x"]).to(dev); y = torch.from_numpy(z["y"].astype(np.int64)).to(dev)
    lg = m(x); ref = torch.from_numpy(z["logits"]); got = lg.detach().cpu()
    flips = got.argmax(1) != ref.argmax(1); margin = (ref[:, 0] - ref[:, 1]).abs()
    print(dt, "logits maxerr %.3e  flips %d  max margin at flips %.3e" % ((got - ref).abs().max(), int(flips.sum()), float(margin[flips].max()) if flips.any() else 0))
    l = L.HybirdLoss()(lg, y); l.backward()
    print("  loss %.7f ref %.7f" % (l.item(), float(z["loss/hybird"])))
    for k, p in m.named_parameters():
        if p.grad is None or ("g/" + k) not in z.files: continue
        r = torch.from_numpy(z["g/" + k]); g = p.grad.cpu()
        print("  %-48s max|ref| %.2e  maxerr/max %.3f  relL2 %.3f" % (k, r.abs().max(), (g - r).abs().max() / r.abs().max().clamp_min(1e-30), (g - r).norm() / r.norm().clamp_min(1e-30)))
