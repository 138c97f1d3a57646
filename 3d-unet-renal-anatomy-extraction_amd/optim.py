"""Fused Adam for the native path: ONE kernel launch updates every parameter tensor of the model.

Drop-in for `torch.optim.Adam(params, lr=..., betas=..., eps=...)` as the training scripts construct it
(reference nb_train_iia.py:18: Adam(model.parameters(), lr=1e-4); weight_decay / amsgrad / maximize are
not used by the reference and are rejected here).  The update rule and the state_dict layout
(`state[p] = {'step', 'exp_avg', 'exp_avg_sq'}`) are torch.optim.Adam's, so checkpoints interchange.
Parameters whose `.grad` is None are skipped, exactly like torch (the never-used skip_conv weights and the
conv biases in front of InstanceNorm).
"""
import ctypes

import torch

import _native as N
from _native import check, ptr, stream

_CHUNK = 16384   # elements per workgroup (multiple of 1024)


class _AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("count", ctypes.c_int64)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("ru3d optim.Adam implements plain Adam (weight_decay=0, amsgrad=False)")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._plans = {}

    def _plan(self, gi, group):
        """Static part of the launch: block map + pinned host table (built once per param group)."""
        plan = self._plans.get(gi)
        params = [p for p in group["params"]]
        if plan is not None and plan["n"] == len(params):
            return plan
        dev = params[0].device
        blocks = []
        for ti, p in enumerate(params):
            N.require_device(p, "parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise N.Ru3dError("optim.Adam: parameters must be contiguous float32")
            for c in range((p.numel() + _CHUNK - 1) // _CHUNK):
                blocks += [ti, c]
        bm = torch.tensor(blocks, dtype=torch.int32).to(dev)
        host = torch.empty(len(params) * ctypes.sizeof(_AdamTensor), dtype=torch.uint8).pin_memory()
        table = torch.empty(host.numel(), dtype=torch.uint8, device=dev)
        plan = {"n": len(params), "block_map": bm, "nblocks": len(blocks) // 2, "host": host, "table": table}
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            plan = self._plan(gi, group)
            N.note_device(params[0].device)
            b1, b2 = group["betas"]
            if plan.get("copied") is not None:
                plan["copied"].synchronize()     # previous step's async H2D of the table has left the host buffer
            arr = (_AdamTensor * len(params)).from_buffer(plan["host"].numpy())
            step_no = None
            any_grad = False
            for i, p in enumerate(params):
                if p.grad is None:
                    arr[i] = _AdamTensor(p.data_ptr(), None, None, None, p.numel())
                    continue
                st = self.state[p]
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                    p.grad = g
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step_no = float(st["step"]) if step_no is None else step_no
                if float(st["step"]) != step_no:
                    raise N.Ru3dError("optim.Adam: parameters of one group must share the step count")
                arr[i] = _AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                     p.numel())
                any_grad = True
            if not any_grad:
                continue
            plan["table"].copy_(plan["host"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            plan["copied"] = ev
            bc1 = 1.0 - b1 ** step_no
            bc2 = 1.0 - b2 ** step_no
            check(N.lib.ru3d_adam_multi(ptr(plan["table"]), ptr(plan["block_map"]), plan["nblocks"], _CHUNK,
                                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), bc1, bc2, 1.0,
                                        stream()), "adam_multi")
        import _ops
        _ops.WEIGHTS_EPOCH[0] += 1      # packed copies of the weights are stale now
        return loss
