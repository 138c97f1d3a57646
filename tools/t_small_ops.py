"""Which torch ops (not libru3d kernels) run inside one config-2 training step, and from where:
    python tools/t_small_ops.py
torch.profiler over one eager step, grouped by op and by the innermost frames of this repository."""
import os, sys, collections
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
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter(); where = collections.defaultdict(collections.Counter)
VIEWS = {"aten::view", "aten::slice", "aten::select", "aten::as_strided", "aten::empty", "aten::empty_strided",
         "aten::empty_like", "aten::permute", "aten::reshape", "aten::_unsafe_view", "aten::unsqueeze", "aten::squeeze",
         "aten::expand", "aten::detach", "aten::alias", "aten::t", "aten::transpose", "aten::narrow", "aten::unbind",
         "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh",
         "aten::resolve_conj", "aten::resolve_neg", "aten::stride", "aten::size", "aten::new_empty", "aten::set_",
         "aten::view_as", "aten::unflatten", "aten::flatten", "aten::chunk", "aten::split", "aten::contiguous"}
for e in ev:
    if not e.name.startswith("aten::") or e.name in VIEWS:
        continue
    par = e.cpu_parent
    nested = False
    while par is not None:
        if par.name.startswith("aten::"):
            nested = True
            break
        par = par.cpu_parent
    if nested:
        continue
    cnt[e.name] += 1
    frames = [f for f in (e.stack or []) if ROOT in f and "tools/" not in f][:2]
    where[e.name][" <- ".join(f.replace(ROOT + "/", "").strip() for f in frames)] += 1
for name, c in cnt.most_common(30):
    print("%-28s %4d" % (name, c))
    for w, k in where[name].most_common(6):
        print("        %3d  %s" % (k, w[:200]))
