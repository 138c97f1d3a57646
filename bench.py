#!/usr/bin/env python3
"""Headline benchmark: train voxels/s of the native 3D U-Net step (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = forward (train mode, Dropout3d on) + HybirdLoss + backward + gradient all-reduce (N > 1)
+ Adam(lr=1e-4), on `ResUnet3D(4, 32, 1, 3)`, bf16 storage / fp32 accumulate, 2 x 128^3 synthetic CT
patches per GPU resident in HBM.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline      the dominant kernel (3x3x3 conv 32->32 on the 2x128^3 grid; forward and stride-1
                dgrad launches of the same kernel), timed live with HIP events recorded on the launch
                stream around each of its launches inside the timed steps.
  cpu_baseline  the CPU oracle (oracle/unet_oracle.py, kind "port") timed on the host on a bounded
                sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK = 8.0e12        # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15  # FLOP/s dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--pools", type=int, default=4)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused = optim.Adam (ru3d_adam_multi); torch = torch.optim.Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off")
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (CPU restatement) train step on a bounded sample: 1 x patch^3 instead of batch x patch^3."""
    from oracle import unet_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("RU3D_CPU_THREADS", "16")))   # the GPU box's CPU share per GPU
    torch.set_num_threads(cores)
    patch = args.patch if args.patch <= 128 else 128
    w = O.init_state_dict(args.pools, args.features, 1, args.classes, seed=0)
    x = O.synth_image((1, 1, patch, patch, patch), 1234)
    y = torch.randint(0, args.classes, (1, patch, patch, patch), generator=torch.Generator().manual_seed(1234))
    state = {}
    t0 = time.perf_counter()
    loss, _, grads = O.train_step(w, x, y, args.pools, {"weight_v": [1, 10, 20][:args.classes]})
    O.adam_step(w, grads, state)
    dt = time.perf_counter() - t0
    return {"value": patch ** 3 / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "1 train step (fwd + HybirdLoss + bwd + Adam) of the same model in fp32 on 1x%d^3 "
                      "(half of the bs=2 workload), torch-CPU oracle, %.1f s" % (patch, dt)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs (one-GPU box): RU3D_DIST_BACKEND=gloo RU3D_ONE_DEVICE=1 runs every rank on cuda:0 over gloo, so
        # the multi-rank code path (GradSync buckets, fused Adam on aliased gradients) is exercised without RCCL
        backend = os.environ.get("RU3D_DIST_BACKEND", "nccl")
        if os.environ.get("RU3D_ONE_DEVICE") == "1":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    import _ops as ops
    import loss as loss_mod
    import network

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = network.ResUnet3D(args.pools, args.features, 1, args.classes).to(dev)
    network.set_compute_dtype(model, dtype)
    model.train(not args.eval_mode)
    sync = None
    if world > 1:
        from parallel import GradSync, broadcast_parameters
        broadcast_parameters(model)
        sync = GradSync(model)
    if args.optimizer == "fused":
        import optim
        opt = optim.Adam(model.parameters(), lr=1e-4)      # same update rule, one launch for the whole model
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    wv = [1, 10, 20][:args.classes] if args.classes <= 3 else None
    criterion = loss_mod.HybirdLoss(weight_v=wv)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    shape = (args.batch, 1, args.patch, args.patch, args.patch)
    x = torch.randn(shape, generator=g, device=dev).clamp_(-2.34, 2.64)
    y = torch.randint(0, args.classes, (args.batch,) + shape[2:], generator=g, device=dev)

    def step():
        logits = model(x)
        loss = criterion(logits, y)
        opt.zero_grad()
        if sync is not None:
            sync.begin_step()
        loss.backward()
        if sync is not None:
            sync.finish_step()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the interpreter's full (generation-2) collection walks every object torch imported - ~90 ms, once, a few steps
    # into the process; park the startup objects in the permanent generation before the timed region
    gc.collect()
    gc.freeze()
    probe = None
    if not args.no_probe:
        probe = ops.Probe(cin=args.features, cout=args.features, k=3, stride=1,
                          extent=(args.patch, args.patch, args.patch))
        ops.set_probe(probe)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.set_probe(None)
    loss_value = float(loss.item())

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    vox_per_step = args.batch * args.patch ** 3
    total = vox_per_step * args.steps * world / dt
    out = {
        "metric": "train voxels/sec (128^3 patch, bs=2 per GPU), whole job",
        "value": total, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "per_gpu": total / world, "final_loss": loss_value,
        "config": {"workload": "config2: ResUnet3D(num_pool=%d, num_features=%d, in=1, out=%d) train step "
                               "(fwd+HybirdLoss+bwd+Adam), %dx1x%d^3 per GPU, dropout %s" %
                               (args.pools, args.features, args.classes, args.batch, args.patch,
                                "off" if args.eval_mode else "on"),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world},
    }
    if probe is not None:
        n_launch, mean_ms = probe.result()
        elem = 2 if dtype == torch.bfloat16 else 4
        v = args.batch * args.patch ** 3
        c = args.features
        alg_bytes = (v * c + v * c + 27 * c * c) * elem          # |in| + |out| + |W|
        alg_flops = 2.0 * v * 27 * c * c
        if n_launch:
            t = mean_ms * 1e-3
            hbm = alg_bytes / t
            mf = alg_flops / t
            # the 32->32 3x3x3 conv has AI = 432 FLOP/B against a ridge of 312 (2.5 PF / 8 TB/s): at the spec peaks the
            # MFMA bound (93 us) is the longer of the two minimum times (HBM: 67 us), and the measured HBM traffic equals
            # the algorithmic bytes, so the launch is priced against the dense bf16 MFMA peak; the HBM view rides along
            # HBM bytes per launch of this kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE in separate runs of tools/kbench.py, gfx950 correction applied by
            # tools/pmc_traffic.py); null when no such measurement exists for this shape.
            traffic = None
            key = "conv3d k3 s1 %d->%d on %dx%d^3" % (c, c, args.batch, args.patch)
            try:
                pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json"))
                if pmc and dtype == torch.bfloat16:
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            out["roofline"] = {"kernel": "conv3d k3 s1 %d->%d on %dx%d^3 (fwd + dgrad launches)" %
                                         (c, c, args.batch, args.patch),
                               "bound": "mfma", "achieved": mf / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": mf / MFMA_BF16_PEAK, "traffic": traffic, "launches": n_launch,
                               "avg_ms": mean_ms, "alg_bytes": alg_bytes, "alg_flops": alg_flops,
                               "hbm_gbps": hbm / 1e9, "hbm_frac": hbm / HBM_PEAK}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args)
        except Exception as e:  # pragma: no cover
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
