import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
DEV = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(3)
for n, dims in [(1, (64, 64, 64)), (2, (128, 128, 128))]:
    xv = torch.randn(n, 32, *dims, generator=g)
    w3 = torch.randn(64, 32, 3, 3, 3, generator=g) * 0.03; w1 = torch.randn(64, 32, 1, 1, 1, generator=g) * 0.1
    b3 = torch.randn(64, generator=g); b1 = torch.randn(64, generator=g)
    x = ops.as_input(xv.to(DEV), BF)
    p3 = ops.pack_weight(w3.to(DEV), N.ROLE_CONV_FWD, BF, 2); p1 = ops.pack_weight(w1.to(DEV), N.ROLE_CONV_FWD, BF, 2)
    y3, mean, scale, y1 = ops.conv_s2_pair_fwd_in(x, p3, b3.to(DEV), p1, b1.to(DEV), 64, None)
    ya = ops.conv_fwd(x, p3, b3.to(DEV), 64, 3, 2)
    yb, m2, s2 = ops.conv_fwd_in(x, p3, b3.to(DEV), 64, 3, 2, None)
    yc = ops.conv_fwd(x, p1, b1.to(DEV), 64, 1, 2)
    print(n, dims, "pair vs conv_fwd equal:", torch.equal(y3, ya), "vs conv_fwd_in:", torch.equal(y3, yb),
          "stats equal:", torch.equal(mean, m2), torch.equal(scale, s2), (mean - m2).abs().max().item(), (scale - s2).abs().max().item(),
          "skip equal:", torch.equal(y1, yc), (y1.float() - yc.float()).abs().max().item())
