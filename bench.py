#!/usr/bin/env python3
"""Headline benchmark: train voxels/s of the native 3D U-Net step (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this script starts N fresh child ranks itself (torch.distributed.run, one process per GPU)
before it touches the GPU, and relays rank 0's JSON line.  Under a launcher (WORLD_SIZE set, the driver's
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) it runs as one rank; --gpus must then
equal WORLD_SIZE.

One step = forward (train mode, Dropout3d on) + HybirdLoss + backward + gradient all-reduce over RCCL (N > 1)
+ Adam(lr=1e-4), on `ResUnet3D(4, 32, 1, 3)`, bf16 storage / fp32 accumulate, 2 x 128^3 synthetic CT patches per GPU
resident in HBM.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline        the 3x3x3 conv 32->32 on the 2x128^3 grid (forward and stride-1 dgrad launches of the same kernel),
                  timed live with HIP events recorded on the launch stream around each launch inside the timed steps.
  roofline_step   the whole step against SURVEY 8(d)'s ALG_FLOPS / ALG_BYTES for the configuration.
  parity          N == 1: the bf16 model's eval-mode argmax mask against the CPU oracle's (same weights, same input):
                  per-class Dice (reference `dice`, loss.py:32-48), flipped voxels, flips outside the margin band;
                  the same for the fp32 parity mode of the HIP path.
  cpu_baseline    the CPU oracle (oracle/unet_oracle.py, kind "port") timed on the host on a bounded sample of the
                  same workload (rank 0, N == 1 only): 1 warm-up + 2 timed train steps at bs=1.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-unet-renal-anatomy-extraction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12        # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15  # FLOP/s dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patch", type=int, nargs="+", default=[128], help="one extent (cube) or three (D H W)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--pools", type=int, default=4)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="fused = optim.Adam (ru3d_adam_multi); torch = torch.optim.Adam")
    ap.add_argument("--checkpoint", action="store_true", help="recompute ResBlock interiors in backward")
    ap.add_argument("--grad-transport", default=os.environ.get("RU3D_GRAD_TRANSPORT", "fp32"),
                    choices=["fp32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-torch-adam", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="graph: replay the step from a hipGraph captured after the warm-up (1 GPU, bf16/fp32, fused "
                         "Adam); eager: issue every launch from Python")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: N fresh child processes (this process never touches the GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def conv_plan(pools, features, classes):
    """(cin, cout, k, stride, level, transposed) of every conv of ResUnet3D (reference network.py:470-565)."""
    widths = [features << i for i in range(pools + 1)]
    convs = [(1, features, 3, 1, 0, False), (features, classes, 1, 1, 0, False)]

    def res(cin, cout, stride, level):
        out = [(cin, cout, 3, stride, level, False), (cout, cout, 3, 1, level, False)]
        if cin != cout or stride != 1:
            out.append((cin, cout, 1, stride, level, False))
        return out

    for lv in range(pools + 1):
        for i in range(max(lv, 1)):
            convs += res(widths[lv], widths[lv], 1, lv)
        if lv < pools:
            convs += res(widths[lv], widths[lv + 1], 2, lv + 1)                 # pool: output on level lv+1
            convs.append((widths[lv + 1], widths[lv], 3, 2, lv + 1, True))      # up: input on level lv+1
            convs += res(2 * widths[lv], widths[lv], 1, lv)                     # decode
    return convs


def alg_work(pools, features, classes, batch, dims, elem):
    """ALG_FLOPS = 3 * sum 2*MACs, ALG_BYTES = 3 * sum (|in| + |out| + |W|) * elem."""
    flops = 0.0
    nbytes = 0.0
    for cin, cout, k, stride, level, transposed in conv_plan(pools, features, classes):
        d = [x >> level for x in dims]
        v = batch * d[0] * d[1] * d[2]            # transposed: input positions; else: output positions
        macs = v * cin * cout * k ** 3
        if transposed:
            vin, vout = v, batch * (2 * d[0] - 1) * (2 * d[1] - 1) * (2 * d[2] - 1)
        else:
            vout, vin = v, v * stride ** 3
        flops += 3 * 2.0 * macs
        nbytes += 3 * (vin * cin + vout * cout + cin * cout * k ** 3) * elem
    return flops, nbytes


def host_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(cores, int(os.environ.get("RU3D_CPU_THREADS", "16")))   # the GPU box's CPU share per GPU


def cpu_baseline(args, dims):
    """Oracle (CPU restatement) train steps on a bounded sample: bs=1 instead of bs=2, 1 warm-up + 2 timed."""
    import torch
    from oracle import unet_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    dims = tuple(min(d, 128) for d in dims)
    w = O.init_state_dict(args.pools, args.features, 1, args.classes, seed=0)
    x = O.synth_image((1, 1) + dims, 1234)
    y = torch.randint(0, args.classes, (1,) + dims, generator=torch.Generator().manual_seed(1234))
    state = {}
    times = []
    for i in range(3):
        t0 = time.perf_counter()
        loss, _, grads = O.train_step(w, x, y, args.pools, {"weight_v": [1, 10, 20][:args.classes]})
        O.adam_step(w, grads, state)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / 2
    vox = dims[0] * dims[1] * dims[2]
    return {"value": vox / dt, "unit": "voxels/s", "cores": cores, "kind": "port", "batch": 1,
            "sample": "1 warm-up + 2 timed train steps (fwd + HybirdLoss + bwd + Adam) of the same model in fp32 at "
                      "bs=1 on 1x%dx%dx%d (half of the bs=2 workload), torch-CPU oracle, %.1f s per step "
                      "(warm-up %.1f s)" % (dims + (dt, times[0]))}


def parity_report(args, model, dims, dev, dtype):
    """Dice between the HIP path's argmax mask and the CPU oracle's, same weights and input (BASELINE metric's
    'Dice vs CPU ref'); eval mode (dropout off).  bs=1 of the bench shape, capped at 128^3 for the CPU side."""
    import torch
    import network
    from oracle import unet_oracle as O
    torch.set_num_threads(host_cores())
    dims = tuple(min(d, 128) for d in dims)
    w = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    x = O.synth_image((1, 1) + dims, 4321)
    with torch.no_grad():
        ref = O.unet_forward(x, w, args.pools)                 # fp32 CPU oracle logits [1, C, D, H, W]
    ref_mask = ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    out = {"input": "1x1x%dx%dx%d synthetic CT patch, weights = the benchmarked model after its timed steps" % dims,
           "oracle": "oracle.unet_forward fp32 on the host"}
    was_training = model.training
    model.eval()
    for name, dt, band in (("bf16" if dtype != torch.float16 else "fp16", dtype, 0.05), ("fp32", torch.float32, 5e-5)):
        if dt == torch.float32 and name != "fp32":
            continue
        network.set_compute_dtype(model, dt)
        with torch.no_grad():
            got = model(x.to(dev)).float().cpu()
        mask = got.argmax(1)
        flips = (mask != ref_mask)
        dices = []
        for c in range(args.classes):
            p = (mask == c).double()
            g = (ref_mask == c).double()
            tp = (p * g).sum()
            fn = ((1 - p) * g).sum()
            fp = (p * (1 - g)).sum()
            dices.append(float((tp + 1e-7) / (tp + 0.5 * fn + 0.5 * fp + 1e-7)))   # loss.py:32-48, alpha=beta=0.5
        out[name] = {"dice_vs_cpu_ref": dices, "argmax_flips": int(flips.sum()), "voxels": int(flips.numel()),
                     "flips_beyond_margin": int((flips & (margin > band)).sum()), "margin_band": band,
                     "max_abs_logit_err": float((got - ref).abs().max())}
        if dt == dtype == torch.float32:
            break
    network.set_compute_dtype(model, dtype)
    model.train(was_training)
    return out


class _StdoutToStderr:
    """gloo and RCCL announce themselves on STDOUT (C++ side); bench.py's stdout is one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = max(world, 1)
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks, or drop the launcher and let bench.py "
              "start them)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    one_device = os.environ.get("RU3D_ONE_DEVICE") == "1"
    # RU3D_DP_REHEARSE=1 at N = 1: the data-parallel machinery of N > 1 (GradSync: buckets in autograd order, gradients born
    # inside them, RCCL all-reduce per bucket on the side stream - with a communicator of one rank -, fused Adam on the
    # aliased buckets, the eager loop) on the one GPU there is: what the N > 1 host path and its extra launches cost
    rehearse = world == 1 and os.environ.get("RU3D_DP_REHEARSE") == "1"
    if rehearse:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(0)
        with _StdoutToStderr():
            dist.init_process_group("gloo", rank=0, world_size=1)
    elif world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed = rendezvous + control plane (store, barrier, the max over ranks of the timing) on gloo;
        # the gradient exchange itself is RCCL through the C ABI (parallel.RcclComm) on a side HIP stream.
        # rehearsal knob (one-GPU box): RU3D_ONE_DEVICE=1 runs every rank on cuda:0 and exchanges gradients over gloo
        # (RCCL refuses two ranks on one device), so bucketing / aliasing / fused Adam run exactly as on N GPUs
        if one_device:
            local = 0
        torch.cuda.set_device(local)
        with _StdoutToStderr():
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    import _ops as ops
    import loss as loss_mod
    import network

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    dims = tuple(args.patch) if len(args.patch) == 3 else (args.patch[0],) * 3
    torch.manual_seed(0)
    model = network.ResUnet3D(args.pools, args.features, 1, args.classes).to(dev)
    network.set_compute_dtype(model, dtype)
    if args.checkpoint:
        network.set_checkpointing(model, True)
    model.train(not args.eval_mode)
    sync = None
    transport = None
    if world > 1 or rehearse:
        from parallel import make_grad_sync, broadcast_parameters
        broadcast_parameters(model)
        want = os.environ.get("RU3D_COMM", "torch" if one_device else "rccl")
        gd = torch.bfloat16 if (args.grad_transport == "bf16" and want == "rccl") else torch.float32
        # every rank votes on the transport before any of them enters RCCL's collective set-up, and again on its outcome
        # (parallel.make_grad_sync): either all ranks exchange over RCCL or all over torch.distributed
        with _StdoutToStderr():
            sync = make_grad_sync(model, transport=want, grad_dtype=gd, exchange=os.environ.get("RU3D_EXCHANGE", "allreduce"))
        transport = sync.transport
        if transport != want and rank == 0:
            print("bench.py: RCCL exchange unavailable on some rank (%s); all ranks use torch.distributed" %
                  sync.fallback_reason, file=sys.stderr, flush=True)
    if args.optimizer == "fused":
        import optim
        opt = optim.Adam(model.parameters(), lr=1e-4)      # same update rule, one launch for the whole model
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    scaler = None
    if dtype == torch.float16:
        import optim as optim_mod
        scaler = optim_mod.LossScaler()
    wv = [1, 10, 20][:args.classes] if args.classes <= 3 else None
    criterion = loss_mod.HybirdLoss(weight_v=wv)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    shape = (args.batch, 1) + dims
    x = torch.randn(shape, generator=g, device=dev).clamp_(-2.34, 2.64)
    y = torch.randint(0, args.classes, (args.batch,) + dims, generator=g, device=dev)

    def make_step(optimizer):
        def step():
            logits = model(x)
            loss = criterion(logits, y)
            optimizer.zero_grad()
            if sync is not None:
                sync.begin_step()
            if scaler is not None:
                scaler.scale(loss).backward()
            else:
                loss.backward()
            if sync is not None:
                sync.finish_step()
            if scaler is not None:
                scaler.step(optimizer)
            else:
                optimizer.step()
            return loss
        return step

    eager_step = make_step(opt)
    step = eager_step
    # --launch auto / graph (N = 1, fused Adam, no loss scaling): the whole step (forward, loss, backward, fused Adam) is
    # captured in a hipGraph after the warm-up and the timed region replays it - the same kernels on the same data, bit
    # for bit (graph.py, tests/test_gpu_graph.py).  Round 2 measured the eager loop within 5 % of host-bound (17.5 ms of
    # enqueueing per 18.4 ms step); every kernel saving since then would otherwise disappear behind the interpreter.
    # The eager figure of the same steps rides along (`ms_per_step_eager`); `--launch eager` times that loop alone.
    use_graph = args.launch in ("auto", "graph") and world == 1 and not rehearse and args.optimizer == "fused"
    gstep = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats(dev)
    if use_graph:
        import graph as graph_mod
        gstep = graph_mod.GraphedTrainStep(model, criterion, opt, warmup=max(1, args.warmup), scaler=scaler)

        def step():
            return gstep(x, y)
        for _ in range(max(1, args.warmup) + 1):      # eager warm-up steps, then capture + first replay
            step()
    else:
        for _ in range(args.warmup):
            step()
    fence()
    peak_mem_capture = torch.cuda.max_memory_allocated(dev)
    # the interpreter's full (generation-2) collection walks every object torch imported - ~90 ms, once, a few steps
    # into the process; park the startup objects in the permanent generation before the timed region
    gc.collect()
    gc.freeze()
    probe = None
    if not args.no_probe and dtype != torch.float32:
        probe = ops.Probe(n=args.batch, cin=args.features, cout=args.features, extent=dims)
        if not use_graph:
            ops.set_probe(probe)
    if not use_graph:
        torch.cuda.reset_peak_memory_stats(dev)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_dt = time.perf_counter() - t0          # the host is done enqueueing here; the device may still be running
    fence()
    dt = time.perf_counter() - t0
    ops.set_probe(None)
    loss_value = float(loss.item())
    # a replay allocates nothing: the step's peak is the one seen while it was captured
    peak_mem = peak_mem_capture if use_graph else torch.cuda.max_memory_allocated(dev)
    eager_ms = eager_host_ms = None
    if use_graph:
        # the same steps issued launch by launch, right behind the timed replays: the eager figure for comparison, and
        # the HIP-event timing of the roofline kernel (events recorded inside a captured graph cannot be read back)
        if scaler is not None:
            gstep.release()          # the loss scaler's state returns to the host for the eager steps
            gstep = None
        if probe is not None:
            ops.set_probe(probe)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eager_step()
        eager_host_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        fence()
        eager_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        ops.set_probe(None)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    vox_per_step = args.batch * dims[0] * dims[1] * dims[2]
    total = vox_per_step * args.steps * world / dt
    cube = dims[0] == dims[1] == dims[2]
    shape_txt = ("%d^3" % dims[0]) if cube else "%dx%dx%d" % dims
    out = {
        "metric": "train voxels/sec (%s patch, bs=%d per GPU), whole job" % (shape_txt, args.batch),
        "value": total, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "per_gpu": total / world, "final_loss": loss_value, "peak_mem_gib": peak_mem / 2 ** 30,
        "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
        "step_launch": "hipGraph replay of the captured step (graph.GraphedTrainStep)" if use_graph else
                       "eager: ~500 launches per step issued from Python",
        "config": {"workload": "ResUnet3D(num_pool=%d, num_features=%d, in=1, out=%d) train step "
                               "(fwd+HybirdLoss+bwd+Adam), %dx1x%s per GPU, dropout %s%s" %
                               (args.pools, args.features, args.classes, args.batch, shape_txt,
                                "off" if args.eval_mode else "on", ", activation checkpointing" if args.checkpoint
                                else ""),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "optimizer": "optim.Adam (fused multi-tensor)" if args.optimizer == "fused" else "torch.optim.Adam"},
    }
    if use_graph:
        out["ms_per_step_eager"] = eager_ms
        out["host_enqueue_ms_per_step_eager"] = eager_host_ms
    if rehearse:
        out["config"]["rehearsal"] = "RU3D_DP_REHEARSE=1: N > 1 host path (GradSync + eager loop) with a communicator of one rank"
    if world > 1 or rehearse:
        out["config"]["grad_exchange"] = ("RCCL all-reduce via ru3d_comm_allreduce, side HIP stream, 64 MiB buckets, "
                                          "%s transport" % args.grad_transport) if transport == "rccl" else \
            "torch.distributed (%s)" % dist.get_backend()
    elem = 4 if dtype == torch.float32 else 2
    sflops, sbytes = alg_work(args.pools, args.features, args.classes, args.batch, dims, elem)
    t_step = dt / args.steps
    peak_f = MFMA_BF16_PEAK if elem == 2 else 157.3e12
    out["roofline_step"] = {"alg_flops": sflops, "alg_bytes": sbytes, "achieved_tflops": sflops / t_step / 1e12,
                            "frac_mfma": sflops / t_step / peak_f, "achieved_gbps": sbytes / t_step / 1e9,
                            "frac_hbm": sbytes / t_step / HBM_PEAK, "peak_tflops": peak_f / 1e12,
                            "bound_ms": 1e3 * max(sflops / peak_f, sbytes / HBM_PEAK), "traffic": None}
    # HBM bytes one step really moves: two PMC passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_step.sh) over this very
    # configuration, committed under profiles/ - the figure is only attached to the shape it was measured on
    try:
        if elem == 2 and cube and dims[0] == 128 and args.batch == 2 and args.features == 32 and args.pools == 4:
            steps_pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_step_traffic.json"))
            if steps_pmc:
                t = json.load(open(os.path.join(ROOT, "profiles", steps_pmc[-1])))
                out["roofline_step"]["traffic"] = t["hbm_bytes_per_step"]
                out["roofline_step"]["traffic_over_alg_bytes"] = t["hbm_bytes_per_step"] / sbytes
                out["roofline_step"]["traffic_hbm_ms"] = 1e3 * t["hbm_bytes_per_step"] / HBM_PEAK
                out["roofline_step"]["traffic_source"] = "profiles/" + steps_pmc[-1]
    except Exception:
        pass
    if probe is not None:
        n_launch, mean_ms = probe.result()
        v = vox_per_step
        c = args.features
        alg_bytes = (v * c + v * c + 27 * c * c) * elem          # |in| + |out| + |W|
        alg_flops = 2.0 * v * 27 * c * c
        if n_launch:
            t = mean_ms * 1e-3
            hbm = alg_bytes / t
            mf = alg_flops / t
            # the 32->32 3x3x3 conv has AI = 432 FLOP/B against a ridge of 312 (2.5 PF / 8 TB/s): at the spec peaks the
            # MFMA bound (93 us) is the longer of the two minimum times (HBM: 67 us), so the launch is priced against
            # the dense bf16 MFMA peak; the HBM view rides along.  `traffic` = HBM bytes per launch of this kernel from
            # the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
            # tools/kbench.py, gfx950 correction applied by tools/pmc_traffic.py); null when there is none for the shape.
            traffic = None
            key = "conv3d k3 s1 %d->%d on %dx%d^3" % (c, c, args.batch, dims[0])
            try:
                pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json"))
                if pmc and elem == 2 and cube:
                    traffic = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
            out["roofline"] = {"kernel": "conv3d k3 s1 %d->%d on %dx%s: every launch of the step's eager twin (forward with "
                                         "fused InstanceNorm statistics, input gradient plain and with fused backward "
                                         "sums), HIP events recorded by the library around the kernel alone" %
                                         (c, c, args.batch, shape_txt),
                               "bound": "mfma", "achieved": mf / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                               "frac": mf / MFMA_BF16_PEAK, "traffic": traffic, "launches": n_launch,
                               "avg_ms": mean_ms, "alg_bytes": alg_bytes, "alg_flops": alg_flops,
                               "hbm_gbps": hbm / 1e9, "hbm_frac": hbm / HBM_PEAK}
    if world == 1 and args.optimizer == "fused" and not args.no_torch_adam:
        # the caller-owned optimizer of the reference scripts (nb_train_iia.py:18) on the same model
        topt = torch.optim.Adam(model.parameters(), lr=1e-4)
        if gstep is not None:
            gstep.release()
        tstep = make_step(topt)
        for _ in range(2):
            tstep()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tstep()
        fence()
        out["ms_per_step_torch_adam"] = 1e3 * (time.perf_counter() - t1) / args.steps
        del topt
    if rank == 0 and world == 1 and not args.no_parity:
        try:
            out["parity"] = parity_report(args, model, dims, dev, dtype)
        except Exception as e:  # pragma: no cover
            out["parity"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, dims)
        except Exception as e:  # pragma: no cover
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or rehearse:
        if sync is not None:
            torch.cuda.synchronize()
            sync.remove()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
