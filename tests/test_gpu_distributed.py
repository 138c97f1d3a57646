"""Multi-rank path with the HIP model, rehearsed on ONE MI355X: fresh child processes share cuda:0.

  * two ranks over gloo: averaged gradient == mean over ranks of the CPU oracle's per-shard gradients, fused Adam on
    the bucket-aliased gradients keeps the ranks bit-identical, Dropout3d masks differ per rank;
  * the RCCL wrapper of the C ABI (ru3d_comm_*) with a world of one rank, and GradSync on its side stream;
  * bench.py --gpus 2 starts its own ranks and reports n_gpus = 2.
Needs a real MI355X: run with `-m gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

from oracle import unet_oracle as O  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "dist_child.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(mode, world, out, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, CHILD, mode, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])


def test_two_ranks_hip_model_mean_of_oracle_shard_gradients(tmp_path):
    out = str(tmp_path)
    _run_ranks("hip_dp", 2, out)
    res = [torch.load(os.path.join(out, "rank%d.pt" % r), weights_only=False) for r in range(2)]
    # broadcast: both ranks start from rank 0's weights
    for k in res[0]["w0"]:
        assert torch.equal(res[0]["w0"][k], res[1]["w0"][k]), k
    assert res[0]["buckets"] > 1 and res[0]["aliased"] and res[1]["aliased"]
    # averaged gradients are the same bits on both ranks
    for k, g0 in res[0]["grads1"].items():
        g1 = res[1]["grads1"][k]
        assert (g0 is None) == (g1 is None), k
        if g0 is not None:
            assert torch.equal(g0, g1), k
    # ... and equal the mean of the oracle's per-shard gradients (float64 oracle; tolerance as in the single-rank
    # parity tests: 5e-3 of the tensor's max)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_child import shard
    w64 = {k: v.double() for k, v in res[0]["w0"].items()}
    shards = []
    for r in range(2):
        x, y = shard(r)
        _, _, g = O.train_step(w64, x.double(), y, 2)
        shards.append(g)
    checked = 0
    for k, got in res[0]["grads1"].items():
        if k not in shards[0]:
            assert got is None, k
            continue
        ref = ((shards[0][k] + shards[1][k]) / 2).float()
        if got is None:     # conv bias in front of InstanceNorm: identically zero gradient
            assert k.endswith(("conv1.bias", "conv2.bias")) and float(ref.abs().max()) < 1e-5, k
            continue
        tol = 5e-3 * max(float(ref.abs().max()), 1e-6) + 1e-7
        assert float((got - ref).abs().max()) <= tol, k
        checked += 1
    assert checked > 20
    # three fused Adam steps on bucket-aliased gradients: ranks stay bit-identical, and the weights moved
    moved = 0
    for k in res[0]["w3"]:
        assert torch.equal(res[0]["w3"][k], res[1]["w3"][k]), k
        moved += int(not torch.equal(res[0]["w3"][k], res[0]["w0"][k]))
    assert moved > 20
    # same torch seed on both ranks, different Dropout3d masks
    assert not torch.equal(res[0]["mask"], res[1]["mask"])
    assert set(res[0]["mask"].unique().tolist()) <= {0.0, 2.0}


def test_sync_batchnorm_two_ranks_equals_one_big_batch(tmp_path):
    """ops.set_bn_sync(): two ranks with two samples each == one process with the four samples (torch modules, fp64):
    outputs, input gradients, running statistics; parameter gradients are per-rank sums that add up to the big batch's."""
    import copy
    import network
    out = str(tmp_path)
    _run_ranks("sync_bn", 2, out)
    res = [torch.load(os.path.join(out, "syncbn%d.pt" % r), weights_only=False) for r in range(2)]
    torch.manual_seed(11)
    blk = network.ResBlock(8, 16, stride=2, norm_op=torch.nn.BatchNorm3d, dropout_op=None)
    init = {k: v for k, v in res[0]["state"].items() if "running" not in k and "num_batches" not in k}
    blk.load_state_dict(init, strict=False)      # the children drew gamma / beta after construction
    blk = blk.double().train()
    x = torch.cat([r["x"] for r in res]).double().requires_grad_(True)
    gout = torch.cat([r["gout"] for r in res]).double()
    z = blk(x)
    z.backward(gout)
    for r in range(2):
        sl = slice(2 * r, 2 * r + 2)
        assert (res[r]["z"].double() - z[sl].detach()).abs().max().item() <= 2e-4
        assert (res[r]["gx"].double() - x.grad[sl]).abs().max().item() <= 2e-3 * max(1.0, x.grad.abs().max().item())
        for k in ("norm.running_mean", "norm.running_var"):
            assert (res[r]["state"][k].double() - blk.state_dict()[k]).abs().max().item() <= 1e-5, k
    assert torch.equal(res[0]["state"]["norm.running_var"], res[1]["state"]["norm.running_var"])
    for k, p in blk.named_parameters():
        if p.grad is None:
            continue
        total = res[0]["grads"][k].double() + res[1]["grads"][k].double()
        if k in ("conv1.bias", "conv2.bias"):     # in front of a batch norm without dropout: analytically zero
            assert total.abs().max().item() <= 1e-3 and p.grad.abs().max().item() < 1e-9, k
            continue
        scale = max(p.grad.abs().max().item(), 1e-3)
        assert (total - p.grad).abs().max().item() <= 5e-3 * scale, k


def test_rccl_wrapper_world_of_one(tmp_path):
    out = str(tmp_path)
    _run_ranks("rccl_w1", 1, out)
    res = torch.load(os.path.join(out, "rccl_w1.pt"), weights_only=False)
    assert res["f32_identity"] and res["bf16_roundtrip"], res
    assert res["gradsync_f32_maxerr"] == 0.0, res
    assert res["gradsync_bf16_maxerr"] == 0.0, res
    assert res["buckets"] > 1
    # reduce-scatter + all-gather entry points, the availability probe, the CU budget knob
    assert res["rs_ag_identity"] and res["available"] == 1
    assert res["cu_default"] >= 64 and res["cu_set"] == 240 and res["cu_reserved"] == res["cu_default"] - 16
    assert res["cu_after_remove"] == res["cu_default"]
    # GradSync(exchange="rs_ag") on the MFMA model: same gradients as the plain run; from the second step on the weight
    # gradients are written straight into the buckets (what is still copied: biases, a few hundred elements)
    for step in range(3):
        assert res["arena_step%d_maxerr" % step] == 0.0, res
    (c0, i0), (c1, i1), (c2, i2) = res["arena_stats"]
    assert i0 == 0 and c0 > 1_000_000
    assert c1 == c2 and c1 < 0.01 * c0 and i1 > 0.99 * c0, res["arena_stats"]
    assert res["arena_buckets"] > 1


def test_bench_gpus2_starts_its_own_ranks():
    env = dict(os.environ, RU3D_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--patch",
           "32", "--pools", "2", "--features", "32", "--classes", "2", "--no-probe"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode("utf-8", "replace")[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 4
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"]
