"""A whole training step - forward, loss, backward, fused Adam - captured ONCE in a hipGraph and replayed.

Why: the step is ~550 kernel launches issued from Python through ctypes; on the benchmark configuration the host needs
18.7 ms to enqueue what the device executes in 19.6 ms (bench.py `host_enqueue_ms_per_step`), so the device idles
wherever the host falls behind (weight packing at the start of a step, the short kernels of the 16^3 / 8^3 levels) and
every kernel speed-up is hidden behind the interpreter.  A replay is one `hipGraphLaunch`.

What makes a captured step the SAME computation as the eager one, step after step:
  * the batch is copied into static input buffers in front of the replay;
  * the Dropout3d draws take their counter offset from a device scalar (`ru3d_dropout3d_scale_dev`): replay k adds
    k * (draws per step), i.e. it draws exactly the masks eager step k would have drawn, and the host counter is kept
    in step so eager steps can follow;
  * the fused Adam reads lr, the bias corrections and grad_scale from device memory (`ru3d_adam_multi_dev`); the 28
    bytes per parameter group are rewritten before each replay, so step counts and LR schedules are honoured;
  * gradients, activations and workspaces live at fixed addresses of the graph's private memory pool.
tests/test_gpu_graph.py holds the bit-equality of graphed and eager training.

fp16 storage (round 4): the dynamic loss scaler lives on the device while a step is captured (`ru3d_amp_state`: the update
kernel skips itself on overflow, `ru3d_amp_update` applies apex's schedule), so the reference's own training mode replays
too.  Not captured (the caller falls back to the eager step): a batch of another shape, the multi-GPU gradient exchange.

    step = GraphedTrainStep(model, criterion, optimizer)
    for x, y in loader:
        loss = step(x, y)          # 0-dim device tensor (a static buffer: read it before the next call)
        step.logits                # the network output of that step, for metrics

`Trainer(..., capture_step=True)` drives its training batches through this.
"""
import torch

import _native as N
import _ops
import optim as _optim

_RING = 8      # pinned host blocks in flight: a replay's scalars stay untouched until its upload has executed


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, warmup=2, scaler=None):
        if not isinstance(optimizer, _optim.Adam):
            raise TypeError("GraphedTrainStep needs optim.Adam (its update kernel reads the per-step scalars from "
                            "device memory)")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.scaler = scaler         # optim.LossScaler (fp16 storage): its state moves to the device for the capture
        self.warmup = max(1, int(warmup))    # the first Adam step creates the moment buffers
        self.graph = None
        self.key = None
        self.replays = 0
        self.eager_steps = 0

    # ------------------------------------------------------------------ eager step (warm-up, other shapes)
    def _eager(self, x, y):
        self.optimizer.zero_grad(set_to_none=True)
        logits = self.model(x)
        loss = self.criterion(logits, y)
        stepped = True
        if self.scaler is None:
            loss.backward()
            self.optimizer.step()
        elif self.scaler._dev is not None:
            stepped = self.scaler.eager_step(self.optimizer, loss)
        else:
            self.scaler.scale(loss).backward()
            stepped = self.scaler.step(self.optimizer)
        # a step the loss scaler skipped (overflow) created no Adam state: it does not count as warm-up
        self.eager_steps += 1 if (stepped or self.graph is not None) else 0
        self.skipped_warmup = getattr(self, "skipped_warmup", 0) + (0 if stepped else 1)
        self.logits = logits.detach()
        return loss.detach()

    def _capture(self, x, y):
        dev = x.device
        groups = len(self.optimizer.param_groups)
        self.x = torch.empty_like(x)
        self.y = torch.empty_like(y)
        self.x.copy_(x)
        self.y.copy_(y)
        # device block rewritten before each replay: [0] dropout counter base (int64) | Adam scalars float32 [groups, 8]
        self.block = torch.zeros(8 + groups * 32, dtype=torch.uint8, device=dev)
        self.drop_base = self.block[:8].view(torch.int64)
        self.hyper = self.block[8:].view(torch.float32).view(groups, 8)
        self.host = [torch.zeros(8 + groups * 32, dtype=torch.uint8).pin_memory() for _ in range(_RING)]
        self.host_ev = [None] * _RING
        self.stream = torch.cuda.Stream(dev)
        self.optimizer.begin_capture(self.hyper)
        if self.scaler is not None:
            self.scaler.begin_capture(self.optimizer, dev)
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        drop0 = _ops._drop_counter[0]
        _ops.DROP_OFFSET_BASE[0] = self.drop_base
        import loss as _loss_mod
        _loss_mod.CAPTURE_SINK[0] = self.label_states = []
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                logits = self.model(self.x)
                loss = self.criterion(logits, self.y)
                if self.scaler is None:
                    loss.backward()
                    self.optimizer.step()
                else:
                    self.scaler.scale(loss).backward()
                    self.scaler.step(self.optimizer)
                self.loss = loss.detach()
                self._static_logits = logits.detach()
        finally:
            _ops.DROP_OFFSET_BASE[0] = None
            _loss_mod.CAPTURE_SINK[0] = None
        self.draws = _ops._drop_counter[0] - drop0      # Dropout3d draws of one step
        _ops._drop_counter[0] = drop0                   # nothing has been drawn yet: the first replay is this step
        self.drop_origin = drop0
        self.graph = g
        self.key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype, dev)

    def __call__(self, x, y):
        key = (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype, x.device)
        if self.graph is None:
            if self.eager_steps < self.warmup:
                return self._eager(x, y)
            self._capture(x, y)
        elif key != self.key:
            return self._eager(x, y)        # e.g. the short last batch of an epoch
        else:
            self.x.copy_(x, non_blocking=True)
            self.y.copy_(y, non_blocking=True)
        # eager steps in between (another shape) moved the dropout counter: replay from wherever it stands
        slot = self.replays % _RING
        if self.host_ev[slot] is not None:
            self.host_ev[slot].synchronize()
        host = self.host[slot]
        host[:8].view(torch.int64)[0] = _ops._drop_counter[0] - self.drop_origin
        self.optimizer.replay_scalars(host[8:].view(torch.float32).view(-1, 8))
        self.block.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.host_ev[slot] = ev
        self.graph.replay()
        if self.label_states:
            # out-of-range labels raise F.one_hot's error for replays too (reference loss.py:27): the loss kernel's count
            # rides to the pinned ring behind the replay and is checked at the caller's next read-back
            import loss as _loss_mod
            for st in self.label_states:
                _loss_mod._note_label_flag(st)
        _ops._drop_counter[0] += self.draws
        _ops.WEIGHTS_EPOCH[0] += 1          # the replay's Adam wrote the parameters
        self.replays += 1
        self.logits = self._static_logits   # the step's network output (metrics); like the loss, valid until the next call
        return self.loss

    def release(self):
        """Drop the graph AND everything that keeps its private memory pool alive; step counts go back into the
        optimizer state.  What points into the pool after a capture: the parameters' .grad tensors (the captured backward
        allocated them), the workspace of the capture stream (_native._WS), the captured Adam launch plans, the static
        input / output buffers - the pool returns to the allocator only when all of them are gone."""
        self.optimizer.sync_captured_steps()
        if self.scaler is not None and self.scaler._dev is not None:
            self.scaler.end_capture()
        self.optimizer._captured = None
        for key in [k for k in self.optimizer._plans if k[1]]:       # (group, captured=True)
            del self.optimizer._plans[key]
        if self.graph is not None:
            self.optimizer.zero_grad(set_to_none=True)
            sid = self.stream.cuda_stream
            # the side stream of the small levels' weight gradients (ops._OnSide) joined the capture: its workspace sits
            # in the pool too, and so do the packed weights of the last pass (ops.prepack)
            side = _ops._SIDE.get(self.stream.device)
            sids = (sid,) if side is None else (sid, side.cuda_stream)
            for key in [k for k in N._WS if k[1] in sids]:
                del N._WS[key]
            _ops._PREPACK.clear()
            for name in ("x", "y", "loss", "logits", "_static_logits", "block", "drop_base", "hyper", "host", "host_ev",
                         "stream", "label_states"):
                if hasattr(self, name):
                    delattr(self, name)
        self.graph = None
        self.key = None
