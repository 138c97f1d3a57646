"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (no distributed code at all, nb_train_iia.py:17); the U-Net uses
InstanceNorm (per-sample statistics), so patch volumes shard over ranks as pure data parallelism with exactly one
exchange step per training step: the average of the parameter gradients.

Design (MI355X: 7 xGMI links per GPU, point-to-point):
  * gradients are packed into a few LARGE flat buckets (default 64 MiB) in the order the autograd engine produces
    them - head -> decoder -> bottleneck -> encoder -> stem; 70 % of the parameters live at the bottleneck and are
    ready mid-backward, so their exchange hides behind the expensive, parameter-light encoder backward;
  * transport "rccl" (default on HIP tensors): each full bucket is all-reduced by `ru3d_comm_allreduce` (the RCCL
    wrapper of the C ABI, include/ru3d.h) on a SIDE HIP stream that waits for the bucket's copy-in; the main stream
    keeps running backward kernels and joins the side stream in `finish_step`, before the optimizer.  The
    communicator is created from a unique-id blob that rank 0 publishes through torch.distributed's key-value
    store (torch.distributed is used for rendezvous only);
  * `grad_dtype=torch.bfloat16` ships bf16 buckets (half the bytes per xGMI link; the mean is taken in fp32 inside
    RCCL's reduction and the result is widened back into the fp32 bucket the gradients alias);
  * transport "torch": `torch.distributed.all_reduce` (gloo on CPU tensors for the world_size-2 tests, or the
    process group's own nccl backend);
  * after the exchange `param.grad` aliases its slice of the fp32 bucket (no copy-out);
  * parameters that never receive a gradient (the unused skip_conv of same-shape ResBlocks, reference
    network.py:403-409) are left out identically on every rank.
"""
import ctypes
import os

import torch
import torch.distributed as dist

_COMM_SEQ = [0]


class RcclComm:
    """Communicator of the C ABI (ru3d_comm_*).  `store`: a torch.distributed Store used once, to ship the id."""

    def __init__(self, world, rank, device, store=None):
        import _native as N
        self.N = N
        self.world, self.rank = world, rank
        self.device = torch.device(device)
        blob = (ctypes.c_char * N.COMM_ID_BYTES)()
        key = "ru3d_comm_id_%d" % _COMM_SEQ[0]
        _COMM_SEQ[0] += 1
        if world > 1:
            if store is None:
                store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                N.check(N.lib.ru3d_comm_unique_id(ctypes.cast(blob, ctypes.c_void_p)), "comm_unique_id")
                store.set(key, bytes(blob.raw))
            else:
                blob.raw = bytes(store.get(key))[:N.COMM_ID_BYTES]
        else:
            N.check(N.lib.ru3d_comm_unique_id(ctypes.cast(blob, ctypes.c_void_p)), "comm_unique_id")
        handle = ctypes.c_void_p()
        N.check(N.lib.ru3d_comm_init(ctypes.byref(handle), ctypes.cast(blob, ctypes.c_void_p), world, rank,
                                     self.device.index if self.device.index is not None else
                                     torch.cuda.current_device()), "comm_init")
        self.handle = handle

    def allreduce(self, flat, average, stream):
        N = self.N
        N.check(N.lib.ru3d_comm_allreduce(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                          N.dtype_code(flat.dtype), 1 if average else 0,
                                          ctypes.c_void_p(stream.cuda_stream)), "comm_allreduce")

    def destroy(self):
        if self.handle is not None and self.handle.value:
            self.N.lib.ru3d_comm_destroy(self.handle)
        self.handle = None


class GradSync:
    def __init__(self, module, bucket_bytes=64 << 20, process_group=None, transport=None, grad_dtype=None, comm=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._buckets = []        # persistent flat fp32 buffers, reused every step
        self._wire = []           # persistent bf16 transport buffers (grad_dtype=bf16)
        self._fill = []           # params packed into the bucket being filled
        self._fill_elems = 0
        self._bucket_idx = 0
        self._inflight = []       # (work | None, flat, [(param, offset, numel)])
        self._active = False
        on_gpu = bool(self.params) and self.params[0].is_cuda
        if transport is None:
            transport = os.environ.get("RU3D_COMM", "rccl" if (on_gpu and dist.get_backend(process_group) != "gloo")
                                       else "torch")
        if transport not in ("rccl", "torch"):
            raise ValueError("GradSync transport must be 'rccl' or 'torch'")
        if transport == "rccl" and not on_gpu:
            raise ValueError("GradSync transport 'rccl' needs parameters on a HIP device")
        self.transport = transport
        if grad_dtype is None:
            grad_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("RU3D_GRAD_TRANSPORT", "fp32")]
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("GradSync grad_dtype must be float32 or bfloat16")
        if grad_dtype == torch.bfloat16 and transport != "rccl":
            raise ValueError("bf16 gradient transport needs the 'rccl' transport")
        self.grad_dtype = grad_dtype
        self.comm = None
        self._owns_comm = False
        self._side = None
        if transport == "rccl":
            dev = self.params[0].device
            self.comm = comm
            if comm is None:
                if process_group is not None:
                    raise ValueError("GradSync(transport='rccl') on a sub-group needs an explicit RcclComm")
                self.comm = RcclComm(self.world, self.rank, dev)
                self._owns_comm = True
            self._side = torch.cuda.Stream(device=dev)
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
            return i, self._buckets[i]
        buf = torch.empty(max(elems, self.bucket_elems), dtype=torch.float32, device=like.device)
        if i < len(self._buckets):
            self._buckets[i] = buf
        else:
            self._buckets.append(buf)
        return i, buf

    def _wire_buf(self, i, elems, device):
        while len(self._wire) <= i:
            self._wire.append(None)
        w = self._wire[i]
        if w is None or w.numel() < elems or w.device != device:
            w = self._wire[i] = torch.empty(max(elems, self.bucket_elems), dtype=torch.bfloat16, device=device)
        return w

    def _launch(self):
        if not self._fill:
            return
        params, self._fill = self._fill, []
        elems, self._fill_elems = self._fill_elems, 0
        idx, full = self._flat(elems, params[0].grad)
        flat = full[:elems]
        layout, off = [], 0
        views = []
        for p in params:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p.grad))
            layout.append((p, off, n))
            off += n
        torch._foreach_copy_(views, [p.grad for p in params])      # copy-in (main stream)
        work = None
        if self.transport == "rccl":
            N = self.comm.N
            main = torch.cuda.current_stream(flat.device)
            self._side.wait_stream(main)
            side = ctypes.c_void_p(self._side.cuda_stream)
            if self.grad_dtype == torch.bfloat16:
                wire = self._wire_buf(idx, elems, flat.device)[:elems]
                N.check(N.lib.ru3d_flat_cast(ctypes.c_void_p(flat.data_ptr()), N.F32, ctypes.c_void_p(wire.data_ptr()),
                                             N.BF16, elems, 1.0, side), "flat_cast")
                self.comm.allreduce(wire, True, self._side)
                N.check(N.lib.ru3d_flat_cast(ctypes.c_void_p(wire.data_ptr()), N.BF16, ctypes.c_void_p(flat.data_ptr()),
                                             N.F32, elems, 1.0, side), "flat_cast")
            else:
                self.comm.allreduce(flat, True, self._side)
        elif self._avg_op is not None:
            work = dist.all_reduce(flat, op=self._avg_op, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, layout))

    def finish_step(self):
        """Flush the last partial bucket, wait for every exchange, alias grads to the reduced buckets."""
        self._launch()
        self._active = False
        if self.transport == "rccl" and self._inflight:
            torch.cuda.current_stream(self._inflight[0][1].device).wait_stream(self._side)
        for work, flat, layout in self._inflight:
            if work is not None:
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
        if self._owns_comm and self.comm is not None:
            self.comm.destroy()
        self.comm = None


def broadcast_parameters(module, src=0, process_group=None):
    """Identical initial weights on every rank (rank `src`'s)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
    try:                                   # `.data` writes do not bump torch's version counters
        import _ops
        _ops.WEIGHTS_EPOCH[0] += 1
    except ImportError:
        pass
