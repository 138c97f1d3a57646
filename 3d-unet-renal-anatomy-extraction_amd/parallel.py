"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (no distributed code at all); the U-Net uses
InstanceNorm (per-sample statistics), so patch volumes shard over ranks as pure data parallelism
with exactly one exchange step per training step: the average of the parameter gradients.

Design (MI355X: 7 xGMI links per GPU, point-to-point):
  * gradients are packed into a few LARGE flat buckets (default 64 MiB fp32) in the order the
    autograd engine produces them - head -> decoder -> bottleneck -> encoder -> stem; 70 % of the
    parameters live at the bottleneck and are ready mid-backward, so their exchange hides behind the
    expensive, parameter-light encoder backward;
  * each full bucket is all-reduced asynchronously (torch.distributed backend "nccl" == RCCL runs it
    on its own HIP stream, ordered after the bucket's copy-in) while the main stream keeps running
    backward kernels; `finish_step` joins the streams before the optimizer;
  * after the exchange `param.grad` simply aliases its slice of the bucket (no copy-out);
  * parameters that never receive a gradient (the unused skip_conv of same-shape ResBlocks,
    reference network.py:403-409) are left out identically on every rank.
Works on CPU tensors with the gloo backend too (world_size-2 tests).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, module, bucket_bytes=64 << 20, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._buckets = []        # persistent flat buffers, reused every step
        self._fill = []           # params packed into the bucket being filled
        self._fill_elems = 0
        self._bucket_idx = 0
        self._inflight = []       # (work, flat, [(param, offset, numel)])
        self._active = False
        backend = dist.get_backend(process_group)
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None

    # -- lifecycle ------------------------------------------------------------------------------
    def begin_step(self):
        self._fill, self._fill_elems, self._bucket_idx = [], 0, 0
        self._inflight = []
        self._active = True

    def _on_grad(self, param):
        if not self._active or param.grad is None:
            return
        self._fill.append(param)
        self._fill_elems += param.numel()
        if self._fill_elems >= self.bucket_elems:
            self._launch()

    def _flat(self, elems, like):
        i = self._bucket_idx
        self._bucket_idx += 1
        if i < len(self._buckets) and self._buckets[i].numel() >= elems and self._buckets[i].device == like.device:
            return self._buckets[i]
        buf = torch.empty(max(elems, self.bucket_elems), dtype=torch.float32, device=like.device)
        if i < len(self._buckets):
            self._buckets[i] = buf
        else:
            self._buckets.append(buf)
        return buf

    def _launch(self):
        if not self._fill:
            return
        params, self._fill = self._fill, []
        elems, self._fill_elems = self._fill_elems, 0
        flat = self._flat(elems, params[0].grad)[:elems]
        layout, off = [], 0
        views = []
        for p in params:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p.grad))
            layout.append((p, off, n))
            off += n
        torch._foreach_copy_(views, [p.grad for p in params])      # copy-in (main stream)
        if self._avg_op is not None:
            work = dist.all_reduce(flat, op=self._avg_op, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, layout))

    def finish_step(self):
        """Flush the last partial bucket, wait for every exchange, alias grads to the reduced buckets."""
        self._launch()
        self._active = False
        for work, flat, layout in self._inflight:
            work.wait()
            if self._avg_op is None:
                flat.div_(self.world)
            for p, off, n in layout:
                p.grad = flat[off:off + n].view_as(p)
        self._inflight = []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def broadcast_parameters(module, src=0, process_group=None):
    """Identical initial weights on every rank (rank `src`'s)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
    try:                                   # `.data` writes do not bump torch's version counters
        import _ops
        _ops.WEIGHTS_EPOCH[0] += 1
    except ImportError:
        pass
