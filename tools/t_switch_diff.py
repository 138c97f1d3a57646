"""Per-parameter gradient differences between the default kernels and a switch setting:
    python tools/t_switch_diff.py in RU3D_CONV_S2=0 RU3D_FUSED_SKIP=0 ..."""
import os, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1]; env = dict(a.split("=") for a in sys.argv[2:])
tmp = tempfile.mkdtemp()
def run(tag, e):
    out = os.path.join(tmp, tag + ".pt")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_child.py"), out, kind], env=dict(os.environ, **e), check=True)
    return torch.load(out, weights_only=False)
a, b, c = run("a", {}), run("b", env), run("c", {})
print("loss", a["loss"], b["loss"], "logits", (a["logits"] - b["logits"]).abs().max().item(), "default twice equal:", all(torch.equal(a["grads"][k], c["grads"][k]) for k in a["grads"]))
rows = []
for k, g in a["grads"].items():
    rows.append(((b["grads"][k] - g).norm().item() / max(g.norm().item(), 1e-30), g.norm().item(), k))
for r in sorted(rows, reverse=True)[:12]:
    print("%.4f  |g| %.3e  %s" % r)
