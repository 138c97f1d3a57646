"""Relative gradient errors (fp32 HIP vs float64 oracle) per parameter for a small config: python tools/diag_cfg.py pools feat classes d h w"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import torch, network, loss as L
from oracle import unet_oracle as O
pools, feat, classes, d, h, w = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
torch.manual_seed(7)
model = network.ResUnet3D(pools, feat, 1, classes).to(dev).eval()
wd = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
x = O.synth_image((1, 1, d, h, w), 11)
y = O.phantom_labels(1, (d, h, w), classes)
logits = model(x.to(dev)); loss = L.HybirdLoss(weight_v=[1, 10, 20])(logits, y.to(dev)); loss.backward()
rl, rlog, g64 = O.train_step(wd, x.double(), y, pools, loss_kwargs={"weight_v": [1, 10, 20]})
# the reference's own arithmetic (fp32 CPU) against the same float64 run
rl32, rlog32, g32 = O.train_step({k: v.float() for k, v in wd.items()}, x, y, pools, loss_kwargs={"weight_v": [1, 10, 20]})
rows = []
for k, p in model.named_parameters():
    if p.grad is None or k not in g64: continue
    ref = g64[k].float(); m = ref.abs().max().item()
    rows.append(((p.grad.cpu() - ref).abs().max().item() / max(m, 1e-30), (g32[k] - ref).abs().max().item() / max(m, 1e-30), m, k))
rows.sort(reverse=True)
print("logits err", (logits.cpu() - rlog.float()).abs().max().item(), "loss", loss.item(), rl.item())
for r in rows[:12]: print("hip %.3e  cpu-fp32 %.3e  max %.3e  %s" % r)
