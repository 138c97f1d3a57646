#!/usr/bin/env python3
"""Golden fixtures of the attention-gated / BatchNorm model variants from the *reference itself*:
reference network.py `ResAttrUnet3D` (:72-101) and `ResAttrBNUnet3D` (:38-69) with reference loss.py `HybirdLoss`, torch
CPU fp32, one forward + backward on the G1-style synthetic 32^3 case.  Only tensors are stored (g8_variants.npz).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_variants.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402  (loads the reference's network.py / loss.py)


def main():
    out = {}
    x = G.synth_image((2, 1, 32, 32, 32), 77)
    y = G.phantom_labels(2, (32, 32, 32), 3)
    out["x"], out["y"] = x.numpy(), y.numpy().astype(np.uint8)
    for tag, ctor, train in (("attr", G.ref_network.ResAttrUnet3D, False), ("attrbn", G.ref_network.ResAttrBNUnet3D, True)):
        torch.manual_seed(0)
        model = ctor(num_pool=2, num_features=8, in_channels=1, out_channels=3)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout3d):
                m.p = 0.0
        model.train(train)
        for k, v in model.state_dict().items():
            out["%s/w/%s" % (tag, k)] = v.detach().numpy().copy()
        logits = model(x)
        loss = G.ref_loss.HybirdLoss(weight_v=[1, 10, 20])(logits, y)
        loss.backward()
        out[tag + "/logits"] = logits.detach().numpy()
        out[tag + "/loss"] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["%s/g/%s" % (tag, k)] = p.grad.numpy().copy()
        for k, v in model.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out["%s/after/%s" % (tag, k)] = v.detach().numpy().copy()
        if train:      # inference with the running statistics the training forward left behind
            model.eval()
            with torch.no_grad():
                out[tag + "/logits_eval"] = model(x).numpy()
        print(tag, float(loss), logits.shape)
    np.savez_compressed(os.path.join(G.OUT, "g8_variants.npz"), **out)
    print("wrote g8_variants.npz")


if __name__ == "__main__":
    main()
