"""Model-level gradient check on a shape where the stride-2 tile kernels / pair kernels engage: the config-2 architecture
on 2 x 64^3, dropout off, bf16 storage, every parameter gradient against the CPU oracle (fp32):
    [RU3D_CONV_S2=0 ...] python tools/t_grad_oracle.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")):
    sys.path.insert(0, p)
import torch
import loss as L, network
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
torch.manual_seed(21)
model = network.ResUnet3D(4, 32, 1, 3).to(dev)
w0 = {k: v.detach().cpu().float().clone() for k, v in model.state_dict().items()}
network.set_compute_dtype(model, torch.bfloat16)
model.eval()
shape = (2, 1, 64, 64, 64)
x = O.synth_image(shape, 5); y = O.phantom_labels(2, shape[2:], 3)
loss = L.HybirdLoss(weight_v=[1, 10, 20])(model(x.to(dev)), y.to(dev))
loss.backward(); torch.cuda.synchronize()
torch.set_num_threads(16)
lo, _, g = O.train_step(w0, x, y, 4, loss_kwargs={"weight_v": [1, 10, 20]})
print("loss gpu %.6f oracle %.6f" % (float(loss), float(lo)))
rows = []
for k, p in model.named_parameters():
    if p.grad is None or k not in g: continue
    ref = g[k].float(); got = p.grad.float().cpu()
    rows.append(((got - ref).norm().item() / max(ref.norm().item(), 1e-30), ref.norm().item(), k))
rows.sort(reverse=True)
for r in rows[:10]: print("%.4f  |g| %.3e  %s" % r)
print("median rel err %.4f over %d tensors" % (sorted(r[0] for r in rows)[len(rows) // 2], len(rows)))
