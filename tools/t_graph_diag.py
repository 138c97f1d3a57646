import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_gpu_graph as G
batches = G._batches(G.STEPS)
lr_at = {4: 5e-4}
m_e, o_e, l_e = G._run_eager(torch.bfloat16, batches, lr_at)
m_e2, o_e2, l_e2 = G._run_eager(torch.bfloat16, batches, lr_at)
m_g, o_g, l_g, step = G._run_graphed(torch.bfloat16, batches, lr_at)
print("losses eager==eager2", l_e == l_e2, "eager==graph", l_e == l_g)
for (k, a), (_, b), (_, c) in zip(m_e.state_dict().items(), m_e2.state_dict().items(), m_g.state_dict().items()):
    ee, eg = torch.equal(a, b), torch.equal(a, c)
    if not (ee and eg):
        print(k, "eager/eager2", ee, (a - b).abs().max().item(), "eager/graph", eg, (a - c).abs().max().item())
print("done")
sd_e, sd_e2, sd_g = o_e.state_dict(), o_e2.state_dict(), o_g.state_dict()
names = [k for k, p in m_e.named_parameters()]
for k in sd_e["state"]:
    for f in ("exp_avg", "exp_avg_sq"):
        a, b, c = sd_e["state"][k][f], sd_e2["state"][k][f], sd_g["state"][k][f]
        if not (torch.equal(a, b) and torch.equal(a, c)):
            print(k, names[k] if k < len(names) else "?", f, "e/e2", torch.equal(a, b), "e/g", torch.equal(a, c), (a - c).abs().max().item(), a.abs().max().item())
