"""Per-step wall time of the first steps of a fresh process (GPU box): python tools/steptimes.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import torch, network, loss as L, optim
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = network.ResUnet3D(4, 32, 1, 3).to(dev)
network.set_compute_dtype(model, torch.bfloat16)
model.train()
opt = optim.Adam(model.parameters(), lr=1e-4)
crit = L.HybirdLoss(weight_v=[1, 10, 20])
x = torch.randn(2, 1, 128, 128, 128, device=dev).clamp_(-2.34, 2.64)
y = torch.randint(0, 3, (2, 128, 128, 128), device=dev)
ts = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = crit(model(x), y); opt.zero_grad(); loss.backward(); opt.step()
    t1 = time.perf_counter()          # host done issuing
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print("step: host-issue ms / total ms")
print(" ".join("%d:%.1f/%.1f" % (i, a, b) for i, (a, b) in enumerate(ts)))
print("reserved MB", torch.cuda.memory_reserved() / 2**20, "allocated peak MB", torch.cuda.max_memory_allocated() / 2**20)
# second pass: split the step and watch allocator / gc counters around the outlier
import gc
gc.collect()
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
st = torch.cuda.memory_stats()
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(x); l = crit(out, y); ta = time.perf_counter()
    opt.zero_grad(); l.backward(); tb = time.perf_counter()
    opt.step(); tc = time.perf_counter()
    torch.cuda.synchronize(); td = time.perf_counter()
    s2 = torch.cuda.memory_stats()
    print("%d fwd %.1f bwd %.1f opt %.1f total %.1f | segs %d allocs_retries %d gc %s" % (
        i, 1e3 * (ta - t0), 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (td - t0),
        s2["segment.all.current"], s2["num_alloc_retries"], gc.get_count()))
