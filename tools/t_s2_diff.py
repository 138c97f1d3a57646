import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
DEV = torch.device("cuda:0"); BF = torch.bfloat16; F = torch.nn.functional
cin, cout, dims = 64, 32, (32, 32, 32)
g = torch.Generator().manual_seed(1)
d, h, w = dims
xv = torch.randn(2, cin, d, h, w, generator=g)
wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * (1.0 / (27 * cin / 8) ** 0.5)
b = torch.randn(cout, generator=g)
x = ops.as_input(xv.to(DEV), BF)
pw = ops.pack_weight(wt.to(DEV), N.ROLE_CONVT_FWD, BF)
for bias in (None, b):
    y = ops.convt_fwd(x, pw, bias.to(DEV) if bias is not None else None, cout).float().cpu()
    ref = F.pad(F.conv_transpose3d(xv.bfloat16().float(), wt.bfloat16().float(), bias, stride=2, padding=1), (0, 1, 0, 1, 0, 1))
    e = (y - ref).abs()
    print("bias" if bias is not None else "nobias", "max err", e.max().item())
    bad = (e > 0.1).nonzero()
    print("bad count", bad.shape[0], "of", e.numel())
    if bad.shape[0]:
        for dim, name in ((1, "c"), (2, "d"), (3, "h"), (4, "w")):
            vals = bad[:, dim].unique()
            print(name, vals[:40].tolist(), "n=", vals.numel())
        i = bad[0]; print("first", i.tolist(), y[tuple(i)].item(), ref[tuple(i)].item())
