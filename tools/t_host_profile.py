"""Where the host time of an eager training step goes (cProfile over 5 steps, config 2): python tools/t_host_profile.py"""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import torch
import network, loss as loss_mod, optim
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = network.ResUnet3D(4, 32, 1, 3).to(dev)
network.set_compute_dtype(model, torch.bfloat16)
model.train()
opt = optim.Adam(model.parameters(), lr=1e-4)
crit = loss_mod.HybirdLoss(weight_v=[1, 10, 20])
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((2, 1, 128, 128, 128), generator=g, device=dev)
y = torch.randint(0, 3, (2, 128, 128, 128), generator=g, device=dev)
def step():
    logits = model(x); l = crit(logits, y); opt.zero_grad(); l.backward(); opt.step(); return l
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.2f ms/step, with device %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28); print(s.getvalue()[:6000])
