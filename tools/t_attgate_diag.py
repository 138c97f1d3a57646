"""dev check: bf16 / fp16 logits of the attention-gate net against its own fp32 run (the yardstick of
tests/test_gpu_variants.py::test_attention_gate_16_bit_and_padded), for A/B runs of kernel switches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd"))
import network
DEV = torch.device("cuda:0")
for feat, dtype in [(32, torch.bfloat16), (30, torch.bfloat16), (30, torch.float16)]:
    for seed in (2, 3, 4):
        torch.manual_seed(seed)
        model = network.ResAttrUnet3D(2, feat, 1, 2).to(DEV).eval()
        x = torch.randn(1, 1, 32, 32, 32, device=DEV)
        with torch.no_grad():
            ref = model(x)
            network.set_compute_dtype(model, dtype)
            logits = model(x)
        d = (logits - ref).float()
        print("F=%d %s seed %d: max %.4f  rel-L2 %.4f" % (feat, dtype, seed, d.abs().max().item(), (d.norm() / ref.norm()).item()), flush=True)
