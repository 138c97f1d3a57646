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
        self._captured = None       # graph.GraphedTrainStep: {"hyper": device float32[groups, 8], "steps": [..]}

    def _plan(self, gi, group, captured=False):
        """Static part of the launch: block map + pinned host table (built once per param group; a captured step keeps
        its own table - its memcpy node re-reads the pinned block at every replay)."""
        plan = self._plans.get((gi, captured))
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
        self._plans[(gi, captured)] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """grad_scale: factor applied to every gradient inside the update kernel (1 / loss_scale in fp16 mode)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = self._captured is not None and torch.cuda.is_current_stream_capturing()
        if self._captured is not None and not capturing:
            self.sync_captured_steps()           # an eager step between replays continues their step count
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            if not params:
                continue
            plan = self._plan(gi, group, capturing)
            N.note_device(params[0].device)
            b1, b2 = group["betas"]
            if plan.get("copied") is not None and not capturing:
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
                    if capturing:
                        raise N.Ru3dError("optim.Adam: take one eager step before capturing (the moment buffers are "
                                          "created and zeroed by the first step)")
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
            if capturing:
                # the captured launch reads lr / bias corrections / grad_scale from the device block that
                # hyper_for_replay() refreshes in front of every replay; the table copy is part of the graph (the
                # gradients live at fixed addresses of the graph's memory pool)
                plan["copied"] = None
                self._captured["steps"][gi] = step_no - 1.0     # the first replay IS this step
                amp = self._captured.get("amp")
                if amp is not None:
                    # fp16: the device-side loss scaler decides (skip on overflow, 1 / scale, the true step number)
                    self._captured["amp_base"][gi] = step_no - 1.0
                    check(N.lib.ru3d_adam_multi_amp(ptr(plan["table"]), ptr(plan["block_map"]), plan["nblocks"], _CHUNK,
                                                    ptr(self._captured["hyper"][gi]), ptr(amp), stream()), "adam_multi_amp")
                else:
                    check(N.lib.ru3d_adam_multi_dev(ptr(plan["table"]), ptr(plan["block_map"]), plan["nblocks"], _CHUNK,
                                                    ptr(self._captured["hyper"][gi]), stream()), "adam_multi_dev")
                continue
            ev = torch.cuda.Event()
            ev.record()
            plan["copied"] = ev
            if self._captured is not None and gi in self._captured["steps"]:
                self._captured["steps"][gi] = step_no
                self._captured["amp_base"][gi] = step_no      # (the scaler re-uploads its block with steps = 0)
            bc1 = 1.0 - b1 ** step_no
            bc2 = 1.0 - b2 ** step_no
            check(N.lib.ru3d_adam_multi(ptr(plan["table"]), ptr(plan["block_map"]), plan["nblocks"], _CHUNK,
                                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), bc1, bc2,
                                        float(grad_scale), stream()), "adam_multi")
        import _ops
        _ops.WEIGHTS_EPOCH[0] += 1      # packed copies of the weights are stale now
        return loss


    # ---- a training step captured in a hipGraph (graph.GraphedTrainStep)
    def begin_capture(self, hyper):
        """hyper: device float32 [len(param_groups), 8]; the captured update kernels read their scalars from it."""
        if hyper.shape != (len(self.param_groups), 8) or hyper.dtype != torch.float32:
            raise ValueError("begin_capture: hyper must be float32 [groups, 8]")
        self._captured = {"hyper": hyper, "steps": {}, "amp": None, "amp_base": {}}
        for gi, group in enumerate(self.param_groups):       # device-side plan pieces cannot be made while capturing
            if group["params"]:
                self._plan(gi, group, True)

    def replay_scalars(self, out, grad_scale=1.0):
        """Advance the step counts of the captured groups by one and write this step's scalars into `out` (CPU float32
        [groups, 8]: lr, beta1, beta2, eps, bias_corr1, bias_corr2, grad_scale, sqrt(bias_corr2)) - the caller uploads them."""
        cap = self._captured
        cap["dirty"] = True
        for gi, step_no in cap["steps"].items():
            step_no += 1.0
            cap["steps"][gi] = step_no
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            row = out[gi]
            row[0] = group["lr"]; row[1] = b1; row[2] = b2; row[3] = group["eps"]
            row[4] = 1.0 - b1 ** step_no; row[5] = 1.0 - b2 ** step_no; row[6] = grad_scale
            row[7] = float(row[5]) ** 0.5     # sqrt of the float32 bias_corr2, as ru3d_adam_multi takes it on the host
            if cap.get("amp") is not None:
                # device-side loss scaler: the kernel derives the step number from the steps really taken (skips are
                # decided on the device); slot 5 carries the count before the capture as an int32
                row[5:6].view(torch.int32)[0] = int(cap["amp_base"].get(gi, 0.0))
                row[4] = b1 - float(torch.tensor(b1, dtype=torch.float32))      # residuals: beta = float32 value + this
                row[7] = b2 - float(torch.tensor(b2, dtype=torch.float32))

    def sync_captured_steps(self):
        """Write the step counts reached by graph replays back into state[p]['step'] (state_dict fidelity)."""
        cap = self._captured
        if cap is None or not cap.get("dirty"):
            return
        cap["dirty"] = False
        if cap.get("amp") is not None:
            # the device counted the steps that were not skipped (one 32-byte read-back, on demand only)
            taken = int(cap["amp"].view(torch.int32)[5].item())
            for gi in list(cap["steps"]):
                cap["steps"][gi] = cap["amp_base"].get(gi, 0.0) + taken
        for gi, step_no in cap["steps"].items():
            for p in self.param_groups[gi]["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = torch.tensor(float(step_no))

    def state_dict(self):
        self.sync_captured_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch's layout; the step counts are kept on the HOST whatever device the checkpoint was mapped to (step() reads
        them every call: a device tensor there is a sync per step and cannot be read inside a graph capture)."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].device.type != "cpu":
                st["step"] = st["step"].detach().to("cpu", torch.float32)


class _GradTable:
    """Device table of (grad pointer, count) per parameter in ru3d_adam_tensor layout + block map, rebuilt when the
    gradient tensors move (GradSync re-aliases them into its buckets; autograd allocates fresh ones otherwise)."""

    def __init__(self):
        self.key = None
        self.table = self.block_map = None
        self.nblocks = 0
        self.host = None
        self.copied = None

    def prepare(self, params, dev):
        """The allocations (pinned host block, device table, block map): not allowed while a stream is capturing, so a
        captured step calls this beforehand (LossScaler.begin_capture)."""
        n = len(params)
        if self.host is not None and self.host.numel() == n * ctypes.sizeof(_AdamTensor):
            return
        self.host = torch.empty(n * ctypes.sizeof(_AdamTensor), dtype=torch.uint8).pin_memory()
        self.table = torch.empty(self.host.numel(), dtype=torch.uint8, device=dev)
        blocks = []
        for ti, p in enumerate(params):
            for c in range((p.numel() + _CHUNK - 1) // _CHUNK):
                blocks += [ti, c]
        self.block_map = torch.tensor(blocks, dtype=torch.int32).to(dev)
        self.nblocks = len(blocks) // 2
        self.key = None

    def update(self, params, capturing=False):
        grads = [p.grad for p in params]
        dev = next(g.device for g in grads if g is not None)
        key = tuple((0 if g is None else g.data_ptr(), p.numel()) for p, g in zip(params, grads))
        if key == self.key:
            return
        if self.copied is not None:
            self.copied.synchronize()
        n = len(params)
        self.prepare(params, dev)
        arr = (_AdamTensor * n).from_buffer(self.host.numpy())
        for i, (p, g) in enumerate(zip(params, grads)):
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                raise N.Ru3dError("LossScaler: gradients must be contiguous float32")
            arr[i] = _AdamTensor(None, None if g is None else g.data_ptr(), None, None, p.numel())
        self.table.copy_(self.host, non_blocking=True)
        if capturing:
            self.copied = None          # the copy is a node of the graph; the pinned block stays as it is
        else:
            self.copied = torch.cuda.Event()
            self.copied.record()
        self.key = key


class LossScaler:
    """Dynamic loss scaling for fp16 storage - the reference's apex O1 behaviour (trainer.py:492-493 `amp.scale_loss`,
    538-542 `amp.initialize(..., opt_level)`): start at 2**16, skip the optimizer step and halve the scale when a
    gradient overflows, double it after `growth_interval` (2000) clean steps, up to 2**24 (apex's max_loss_scale).

        scaler = optim.LossScaler()
        scaler.scale(loss).backward()
        scaler.step(optimizer)        # unscale + inf/nan check on the device, one 4-byte read-back, step or skip

    With optim.Adam the 1/scale factor is applied inside the fused update kernel; any other torch optimizer gets its
    gradients unscaled in place first.  bf16 storage needs none of this."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 min_scale=2.0 ** -24, max_scale=2.0 ** 24):
        self.loss_scale = float(init_scale)
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval = int(growth_interval)
        self.min_scale = float(min_scale)
        self.max_scale = float(max_scale)      # apex: max_loss_scale = 2.**24
        self.growth_tracker = 0
        self.skipped_steps = 0
        self._scale_t = None
        self._found = None
        self._tables = {}
        self._dev = None            # captured mode: the ru3d_amp_state block (uint8[32]) on the device
        self._dev_skipped = 0       # its `skipped` count at the last sync

    # ---- captured mode (graph.GraphedTrainStep): the scaler state lives on the device, see ru3d_amp_state
    def begin_capture(self, optimizer, device):
        """Move the state into a device block and tell optim.Adam to take its decisions from there.  Call after
        optimizer.begin_capture()."""
        if not isinstance(optimizer, Adam) or optimizer._captured is None:
            raise TypeError("LossScaler.begin_capture needs an optim.Adam that is being captured")
        self._dev = torch.zeros(32, dtype=torch.uint8, device=device)
        self._upload()
        for gi, group in enumerate(optimizer.param_groups):
            if group["params"]:
                self._tables.setdefault((id(optimizer), gi, "cap"), _GradTable()).prepare(list(group["params"]), device)
        self._scale_t = self._dev[0:4].view(torch.float32).view(())
        self._found = self._dev[8:12].view(torch.float32)
        optimizer._captured["amp"] = self._dev

    def _upload(self):
        f = torch.zeros(8, dtype=torch.float32)
        i = f.view(torch.int32)
        f[0], f[1], f[2] = self.loss_scale, 1.0 / self.loss_scale, 0.0
        i[3], i[4] = self.growth_tracker, 0
        i[5] = 0
        self._dev.copy_(f.view(torch.uint8))
        self._dev_skipped = 0

    def sync(self):
        """Captured mode: read the device block back into the host-side fields (one 32-byte copy; state_dict, logging)."""
        if self._dev is None:
            return
        f = self._dev.cpu().view(torch.float32)
        i = f.view(torch.int32)
        self.loss_scale = float(f[0])
        self.growth_tracker = int(i[3])
        self.skipped_steps += int(i[4]) - self._dev_skipped
        self._dev_skipped = int(i[4])

    def eager_step(self, optimizer, loss):
        """One eager step (a batch of another shape) between replays: the state comes back to the host for it and returns
        to the device block afterwards."""
        dev = self._dev
        self.sync()
        optimizer.sync_captured_steps()
        self._dev = None
        self._scale_t = self._found = None
        try:
            self.scale(loss).backward()
            return self.step(optimizer)
        finally:
            self._dev = dev
            self._upload()
            self._scale_t = dev[0:4].view(torch.float32).view(())
            self._found = dev[8:12].view(torch.float32)

    def end_capture(self):
        self.sync()
        self._dev = None
        self._scale_t = None
        self._found = None

    def scale(self, loss):
        if self._scale_t is None or self._scale_t.device != loss.device:
            self._scale_t = torch.full((), self.loss_scale, dtype=torch.float32, device=loss.device)
            self._found = torch.zeros(1, dtype=torch.float32, device=loss.device)
        return loss * self._scale_t

    def step(self, optimizer):
        """Returns True when the optimizer stepped, False when the step was skipped because of an overflow."""
        if self._scale_t is None:
            raise RuntimeError("LossScaler.step() before LossScaler.scale(loss).backward()")
        fused = isinstance(optimizer, Adam)
        if self._dev is not None:
            if not torch.cuda.is_current_stream_capturing():
                raise RuntimeError("LossScaler: an eager step while the scaler is in captured mode (end_capture() first)")
            # captured: check -> Adam (skips itself on overflow) -> scaler update, all decided on the device
            for gi, group in enumerate(optimizer.param_groups):
                params = [p for p in group["params"]]
                if not any(p.grad is not None for p in params):
                    continue
                tab = self._tables.setdefault((id(optimizer), gi, "cap"), _GradTable())
                tab.key = None
                tab.update(params, capturing=True)
                N.note_device(tab.table.device)
                check(N.lib.ru3d_grad_scale_check(ptr(tab.table), ptr(tab.block_map), tab.nblocks, _CHUNK, 1.0,
                                                  ptr(self._found), stream()), "grad_scale_check")
            optimizer.step()
            check(N.lib.ru3d_amp_update(ptr(self._dev), self.growth_factor, self.backoff_factor, self.growth_interval,
                                        self.min_scale, self.max_scale, stream()), "amp_update")
            return None
        inv = 1.0 / self.loss_scale
        self._found.zero_()
        any_grad = False
        for gi, group in enumerate(optimizer.param_groups):
            params = [p for p in group["params"]]
            if not any(p.grad is not None for p in params):
                continue
            any_grad = True
            tab = self._tables.setdefault((id(optimizer), gi), _GradTable())
            tab.update(params)
            N.note_device(tab.table.device)
            check(N.lib.ru3d_grad_scale_check(ptr(tab.table), ptr(tab.block_map), tab.nblocks, _CHUNK,
                                              1.0 if fused else inv, ptr(self._found), stream()), "grad_scale_check")
        overflow = any_grad and bool(self._found.item() != 0.0)      # the one host read-back of the fp16 step
        if overflow:
            self.loss_scale = max(self.loss_scale * self.backoff_factor, self.min_scale)
            self.growth_tracker = 0
            self.skipped_steps += 1
        else:
            if fused:
                optimizer.step(grad_scale=inv)
            else:
                optimizer.step()
            self.growth_tracker += 1
            if self.growth_tracker >= self.growth_interval:
                self.loss_scale = min(self.loss_scale * self.growth_factor, self.max_scale)
                self.growth_tracker = 0
        self._scale_t.fill_(self.loss_scale)
        return not overflow

    def state_dict(self):
        """apex's `amp.state_dict()` layout, which the reference checkpoints as 'amp_state_dict' (trainer.py:617-618):
        {'loss_scaler0': {'loss_scale': float, 'unskipped': int}} - `unskipped` is apex's name for the count of clean
        steps since the last change of the scale.  The extra key 'ru3d' (skipped-step count, dtype tag) is ignored by
        apex's loader, which reads only the loss_scaler<i> entries."""
        self.sync()
        return {"loss_scaler0": {"loss_scale": self.loss_scale, "unskipped": self.growth_tracker},
                "ru3d": {"dtype": "fp16", "skipped_steps": self.skipped_steps}}

    def load_state_dict(self, state):
        """Reads the apex layout (a checkpoint of the reference) and this class's round-2 flat layout."""
        if not isinstance(state, dict):
            return
        if isinstance(state.get("loss_scaler0"), dict):
            inner = state["loss_scaler0"]
            self.loss_scale = float(inner["loss_scale"])
            self.growth_tracker = int(inner.get("unskipped", 0))
            extra = state.get("ru3d")
            self.skipped_steps = int(extra.get("skipped_steps", 0)) if isinstance(extra, dict) else 0
        elif "loss_scale" in state:
            self.loss_scale = float(state["loss_scale"])
            self.growth_tracker = int(state.get("growth_tracker", 0))
            self.skipped_steps = int(state.get("skipped_steps", 0))
        else:
            return
        self.loss_scale = min(max(self.loss_scale, self.min_scale), self.max_scale)
        if self._scale_t is not None:
            self._scale_t.fill_(self.loss_scale)
