"""Child rank of the multi-process GPU tests (tests/test_gpu_distributed.py starts N of these as fresh processes).

    python dist_child.py <mode> <out_dir>        with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env

modes
  hip_dp     the HIP config-1 model (fp32 parity mode) under data parallelism on ONE device over gloo (RCCL refuses
             two ranks on one GPU): rank-different initial weights -> broadcast, one averaged backward, three fused
             Adam steps on the bucket-aliased gradients, one Dropout3d mask draw.
  sync_bn    a BatchNorm ResBlock with the pooled sums all-reduced over two ranks (SyncBN).
  rccl_w1    the RCCL wrapper of the C ABI with a world of one rank: communicator from a unique id, fp32 / bf16
             all-reduce on a side stream, GradSync(transport="rccl") around the HIP model.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def shard(rank):
    from oracle import unet_oracle as O
    x = O.synth_image((1, 1, 32, 32, 32), 100 + rank)
    y = O.phantom_labels(1, (32, 32, 32), 2)
    if rank % 2:
        y = y.flip(1)
    return x, y


def hip_dp(out):
    import loss as L
    import network
    import optim
    import _ops as ops
    from parallel import GradSync, broadcast_parameters
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    torch.manual_seed(rank)                     # different init per rank: the broadcast must fix it
    model = network.ResUnet3D(2, 8, 1, 2).to(dev)
    broadcast_parameters(model)
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()                                # dropout off: the gradient is a function of (weights, shard) only
    sync = GradSync(model, bucket_bytes=256 << 10, transport="torch")   # several buckets on this small model
    opt = optim.Adam(model.parameters(), lr=1e-4)
    x, y = shard(rank)
    x, y = x.to(dev), y.to(dev)
    crit = L.HybirdLoss()
    grads1 = None
    for step in range(3):
        opt.zero_grad()
        sync.begin_step()
        crit(model(x), y).backward()
        sync.finish_step()
        if step == 0:
            grads1 = {k: (None if p.grad is None else p.grad.detach().cpu().clone())
                      for k, p in model.named_parameters()}
            aliased = all(p.grad is None or any(p.grad.data_ptr() >= b.data_ptr() and
                                                p.grad.data_ptr() < b.data_ptr() + b.numel() * 4
                                                for b in sync._buckets) for p in model.parameters())
        opt.step()
    torch.manual_seed(0)                        # same seed on every rank: the masks must still differ
    mask = ops.dropout_scale(4, 64, 0.5, dev).cpu()
    torch.cuda.synchronize()
    torch.save({"w0": w0, "grads1": grads1, "w3": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "mask": mask, "buckets": len(sync._buckets), "aliased": aliased},
               os.path.join(out, "rank%d.pt" % rank))
    sync.remove()


def rccl_w1(out):
    import loss as L
    import network
    import _native as N
    from parallel import GradSync, RcclComm
    dev = torch.device("cuda:0")
    comm = RcclComm(1, 0, dev)
    side = torch.cuda.Stream(device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    a = torch.randn(1 << 20, generator=g, device=dev)
    ref = a.clone()
    side.wait_stream(torch.cuda.current_stream())
    comm.allreduce(a, True, side)
    b16 = torch.empty(1 << 20, dtype=torch.bfloat16, device=dev)
    import ctypes
    N.check(N.lib.ru3d_flat_cast(ctypes.c_void_p(ref.data_ptr()), N.F32, ctypes.c_void_p(b16.data_ptr()), N.BF16,
                                 ref.numel(), 1.0, ctypes.c_void_p(side.cuda_stream)), "flat_cast")
    comm.allreduce(b16, True, side)
    back = torch.empty_like(ref)
    N.check(N.lib.ru3d_flat_cast(ctypes.c_void_p(b16.data_ptr()), N.BF16, ctypes.c_void_p(back.data_ptr()), N.F32,
                                 ref.numel(), 1.0, ctypes.c_void_p(side.cuda_stream)), "flat_cast")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res = {"f32_identity": bool(torch.equal(a, ref)),
           "bf16_roundtrip": bool(torch.equal(back, ref.to(torch.bfloat16).float()))}
    # the exchange as its two halves, in place on the bucket (world of one: both are the identity)
    c = ref.clone()
    side.wait_stream(torch.cuda.current_stream())
    comm.reduce_scatter(c, c.numel(), True, side)
    comm.all_gather(c, c.numel(), side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res["rs_ag_identity"] = bool(torch.equal(c, ref))
    res["available"] = int(N.lib.ru3d_comm_available())
    comm.destroy()
    # CU budget of the persistent kernels
    res["cu_default"] = int(N.lib.ru3d_get_cu_budget())
    N.check(N.lib.ru3d_set_cu_budget(240), "set_cu_budget")
    res["cu_set"] = int(N.lib.ru3d_get_cu_budget())
    N.check(N.lib.ru3d_set_cu_budget(0), "set_cu_budget")

    torch.manual_seed(0)
    model = network.ResUnet3D(2, 8, 1, 2).to(dev)
    model.eval()
    x, y = shard(0)
    x, y = x.to(dev), y.to(dev)
    L.HybirdLoss()(model(x), y).backward()
    plain = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    for dt in (torch.float32, torch.bfloat16):
        sync = GradSync(model, bucket_bytes=128 << 10, transport="rccl", grad_dtype=dt)
        model.zero_grad(set_to_none=True)
        sync.begin_step()
        L.HybirdLoss()(model(x), y).backward()
        sync.finish_step()
        torch.cuda.synchronize()
        worst = 0.0
        for k, p in model.named_parameters():
            if p.grad is None:
                assert k not in plain
                continue
            want = plain[k] if dt == torch.float32 else plain[k].to(torch.bfloat16).float()
            worst = max(worst, float((p.grad - want).abs().max()))
        res["gradsync_%s_maxerr" % ("f32" if dt == torch.float32 else "bf16")] = worst
        res["buckets"] = len(sync._buckets)
        sync.remove()
    # the MFMA model (bf16 storage), reduce-scatter + all-gather exchange, 16 CUs left to RCCL: from the second step on
    # the weight gradients are born inside the buckets (no copy-in), and they equal the plain run's bit for bit
    torch.manual_seed(0)
    model = network.ResUnet3D(2, 32, 1, 2).to(dev)
    network.set_compute_dtype(model, torch.bfloat16)
    model.eval()
    L.HybirdLoss()(model(x), y).backward()
    plain = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    sync = GradSync(model, bucket_bytes=1 << 20, transport="rccl", exchange="rs_ag", reserve_cus=16)
    res["cu_reserved"] = int(N.lib.ru3d_get_cu_budget())
    stats = []
    for step in range(3):
        model.zero_grad(set_to_none=True)
        sync.begin_step()
        L.HybirdLoss()(model(x), y).backward()
        sync.finish_step()
        torch.cuda.synchronize()
        stats.append((sync.copied_elems, sync.inplace_elems))
        worst = max(float((p.grad - plain[k]).abs().max()) for k, p in model.named_parameters() if p.grad is not None)
        res["arena_step%d_maxerr" % step] = worst
    res["arena_stats"] = stats
    res["arena_buckets"] = len(sync._buckets)
    sync.remove()
    res["cu_after_remove"] = int(N.lib.ru3d_get_cu_budget())
    torch.save(res, os.path.join(out, "rccl_w1.pt"))


def sync_bn(out):
    """A BatchNorm ResBlock (fp32 storage) in training mode with ops.set_bn_sync(): every rank normalises with the
    statistics of the WHOLE batch (both directions pool their sums across ranks)."""
    import network
    import _ops as ops
    rank = dist.get_rank()
    dev = torch.device("cuda:0")
    torch.manual_seed(11)                       # same block on every rank
    blk = network.ResBlock(8, 16, stride=2, norm_op=torch.nn.BatchNorm3d, dropout_op=None).to(dev).train()
    with torch.no_grad():
        blk.norm.weight.uniform_(0.5, 1.5)
        blk.norm.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(40 + rank)
    x = torch.randn(2, 8, 12, 8, 8, generator=g).to(dev).requires_grad_(True)
    gout = torch.randn(2, 16, 6, 4, 4, generator=g).to(dev)
    ops.set_bn_sync()
    z = blk(x)
    z.backward(gout)
    ops.set_bn_sync(enable=False)
    torch.cuda.synchronize()
    torch.save({"x": x.detach().cpu(), "gout": gout.cpu(), "z": z.detach().cpu(), "gx": x.grad.cpu(),
                "state": {k: v.detach().cpu() for k, v in blk.state_dict().items()},
                "grads": {k: p.grad.cpu() for k, p in blk.named_parameters() if p.grad is not None}},
               os.path.join(out, "syncbn%d.pt" % rank))


def main():
    mode, out = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    try:
        {"hip_dp": hip_dp, "rccl_w1": rccl_w1, "sync_bn": sync_bn}[mode](out)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
