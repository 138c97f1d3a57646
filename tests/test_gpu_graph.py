"""graph.GraphedTrainStep: a training step captured in a hipGraph and replayed is the SAME training run as the eager
steps - same losses, same parameters, same Dropout3d masks, same Adam step counts / learning-rate schedule - bit for
bit.  Needs a real MI355X: run with `-m gpu`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import _ops as ops  # noqa: E402
import graph  # noqa: E402
import loss as L  # noqa: E402
import network  # noqa: E402
import optim  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")
STEPS = 7


def _batches(n, shape=(2, 1, 32, 32, 32)):
    out = []
    for i in range(n):
        x = O.synth_image(shape, 900 + i).to(DEV)
        y = O.phantom_labels(shape[0], shape[2:], 3).to(DEV)
        if i % 2:
            y = y.flip(1)
        out.append((x, y))
    return out


def _setup(dtype, seed=3):
    torch.manual_seed(seed)
    model = network.ResUnet3D(2, 32, 1, 3).to(DEV)       # 32 features: the MFMA kernels, Dropout3d p = 0.2 on
    network.set_compute_dtype(model, dtype)
    model.train()
    opt = optim.Adam(model.parameters(), lr=1e-3)
    ops._drop_counter[0] = 0
    return model, opt, L.HybirdLoss()


def _run_eager(dtype, batches, lr_at=None):
    model, opt, crit = _setup(dtype)
    losses = []
    for i, (x, y) in enumerate(batches):
        if lr_at and i in lr_at:
            opt.param_groups[0]["lr"] = lr_at[i]
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return model, opt, [float(v) for v in losses]


def _run_graphed(dtype, batches, lr_at=None, warmup=2):
    model, opt, crit = _setup(dtype)
    step = graph.GraphedTrainStep(model, crit, opt, warmup=warmup)
    losses = []
    for i, (x, y) in enumerate(batches):
        if lr_at and i in lr_at:
            opt.param_groups[0]["lr"] = lr_at[i]
        losses.append(step(x, y).clone())
    torch.cuda.synchronize()
    return model, opt, [float(v) for v in losses], step


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_graphed_training_equals_eager(dtype):
    batches = _batches(STEPS)
    lr_at = {4: 5e-4}                                    # a schedule change between replays
    m_e, o_e, l_e = _run_eager(dtype, batches, lr_at)
    m_g, o_g, l_g, step = _run_graphed(dtype, batches, lr_at)
    assert step.replays == STEPS - 2 and step.eager_steps == 2
    # bf16 storage (the production kernels) is deterministic run to run: bit equality.  The fp32 parity-mode kernels
    # reduce bias / head gradients with float atomics - two eager runs differ in the last bit just the same.
    if dtype == torch.float32:
        def same(a, b):
            return torch.allclose(a, b, rtol=2e-4, atol=1e-7)
        assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(l_g, l_e)), (l_g, l_e)
    else:
        same = torch.equal
        assert l_g == l_e, (l_g, l_e)
    bad = [k for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()) if not same(a, b)]
    assert not bad, bad
    # the optimizer state interchanges with torch.optim.Adam's: step counts follow the replays
    sd_e, sd_g = o_e.state_dict(), o_g.state_dict()
    for k in sd_e["state"]:
        assert float(sd_e["state"][k]["step"]) == float(sd_g["state"][k]["step"]) == STEPS
        assert same(sd_e["state"][k]["exp_avg"], sd_g["state"][k]["exp_avg"])
        assert same(sd_e["state"][k]["exp_avg_sq"], sd_g["state"][k]["exp_avg_sq"])
    assert ops._drop_counter[0] == STEPS * step.draws > 0      # dropout is on and the host counter kept in step


def test_graphed_batchnorm_training_equals_eager():
    """The BatchNorm variant trains on the native kernels (ops.ResBlockBNFn / UpBNFn), so its step is capturable too:
    the running averages and num_batches_tracked are device tensors that every replay moves.  Five steps graphed == the
    same five steps eager: losses, weights, running statistics, batch counts."""
    def run(graphed):
        torch.manual_seed(4)
        model = network.ResAttrBNUnet3D(2, 32, 1, 3).to(DEV)
        network.set_compute_dtype(model, torch.bfloat16)
        model.train()
        opt = optim.Adam(model.parameters(), lr=1e-3)
        ops._drop_counter[0] = 0
        crit = L.HybirdLoss()
        losses = []
        step = graph.GraphedTrainStep(model, crit, opt, warmup=2) if graphed else None
        for x, y in _batches(5):
            if graphed:
                losses.append(float(step(x, y)))
            else:
                opt.zero_grad(set_to_none=True)
                loss = crit(model(x), y)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        if graphed:
            assert step.replays == 3
            step.release()
        return model, losses
    m_e, l_e = run(False)
    m_g, l_g = run(True)
    assert l_e == l_g, (l_e, l_g)
    bad = [k for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()) if not torch.equal(a, b)]
    assert not bad, bad
    tracked = [v for k, v in m_g.state_dict().items() if k.endswith("num_batches_tracked")]
    assert tracked and all(int(v) in (5, 10) for v in tracked)       # ResBlocks use their one norm twice per forward
    moved = [k for k, v in m_g.state_dict().items() if k.endswith("running_mean") and float(v.abs().max()) > 0]
    assert len(moved) >= 8


def test_other_shape_falls_back_and_replays_continue():
    """A batch of another shape (an epoch's short last batch) runs eagerly between replays; the sequence still equals
    the all-eager run, and eager evaluation after the replays sees the updated weights."""
    batches = _batches(5)
    odd = (batches[3][0][:1].contiguous(), batches[3][1][:1].contiguous())
    seq = batches[:3] + [odd] + batches[4:]
    m_e, _, l_e = _run_eager(torch.bfloat16, seq)
    m_g, _, l_g, step = _run_graphed(torch.bfloat16, seq)
    assert step.eager_steps == 3 and step.replays == 2
    assert l_g == l_e, (l_g, l_e)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k
    m_e.eval()
    m_g.eval()
    with torch.no_grad():
        assert torch.equal(m_e(batches[0][0]), m_g(batches[0][0]))
    step.release()
    assert step.graph is None


def test_release_returns_the_graph_pool():
    """GraphedTrainStep.release(): the gradients, the capture stream's workspace, the captured Adam plans and the static
    buffers all point into the graph's private pool - after release() the allocator is back at what the model, the
    optimizer state and the batch need (eager steps continue to work)."""
    import gc
    import graph as G
    import _native as N
    model, opt, crit = _setup(torch.bfloat16)
    (x, y), = _batches(1)
    step = G.GraphedTrainStep(model, crit, opt, warmup=1)
    step(x, y)                                   # eager: Adam moments exist from here on
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    N._WS.clear()
    gc.collect()
    base = torch.cuda.memory_allocated(DEV)
    step(x, y)                                   # capture + first replay
    step(x, y)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated(DEV)
    assert held > base + (8 << 20)               # the pool: activations, gradients, workspaces of one step
    step.release()
    gc.collect()
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated(DEV)
    assert after <= base + (1 << 20), (base, held, after)
    assert not [k for k in opt._plans if k[1]] and all(p.grad is None for p in model.parameters())
    step(x, y)                                   # a released step captures again on demand
    assert step.graph is not None
    step.release()


def test_trainer_capture_step_equals_eager_fit():
    """Trainer.fit(capture_step=True): the epoch results and the final weights of the eager loop."""
    import numpy as np
    import trainer as T

    class Cases(torch.utils.data.Dataset):
        def __init__(self, n=5):
            self.items = [{"image": O.synth_image((1, 1, 32, 32, 32), 700 + i)[0],
                           "label": O.phantom_labels(1, (32, 32, 32), 3)[0]} for i in range(n)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    def fit(capture):
        model, opt, crit = _setup(torch.bfloat16)
        torch.manual_seed(5)
        np.random.seed(5)
        tr = T.Trainer(model=model, optimizer=opt, loss=crit, dataset=Cases(), batch_size=2, valid_split=0.0,
                       dataloader_kwargs={"num_workers": 0}, metrics={"dice": L.Dice()}, progress=False,
                       capture_step=capture)
        tr.fit(num_epochs=3)
        torch.cuda.synchronize()
        return tr, model

    tr_e, m_e = fit(False)
    assert tr_e._graphed is None
    for capture in (True, None):          # None = the default: captured because optimizer and loss are this package's own
        tr_g, m_g = fit(capture)
        assert tr_g._graphed is None                      # fit() releases the graph and its private pool when it returns
        assert tr_g.graph_stats["replays"] >= 3           # 3 epochs x (2 full batches + 1 short one)
        assert tr_g.graph_stats["eager_steps"] >= 3
        assert tr_e.best_result == tr_g.best_result
        for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
            assert torch.equal(a, b), k


def test_trainer_default_keeps_the_eager_loop_for_foreign_optimizer_or_loss():
    """capture_step=None: a caller-owned torch.optim.Adam, or a loss module that is not one of the package's fused
    losses (it may have host-side effects a replay would skip), runs every step from Python as the reference does."""
    import trainer as T

    class Cases(torch.utils.data.Dataset):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            return {"image": O.synth_image((1, 1, 32, 32, 32), 800 + i)[0], "label": O.phantom_labels(1, (32, 32, 32), 3)[0]}

    class Counting(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inner, self.calls = L.HybirdLoss(), 0

        def forward(self, p, t):
            self.calls += 1
            return self.inner(p, t)

    model, opt, _ = _setup(torch.bfloat16)
    crit = Counting()
    tr = T.Trainer(model=model, optimizer=opt, loss=crit, dataset=Cases(), batch_size=1, valid_split=0.0,
                   dataloader_kwargs={"num_workers": 0}, progress=False)
    tr.fit(num_epochs=4)
    assert tr._graphed is None and crit.calls == 8
    model, _, crit = _setup(torch.bfloat16)
    tr = T.Trainer(model=model, optimizer=torch.optim.Adam(model.parameters(), lr=1e-4), loss=crit, dataset=Cases(),
                   batch_size=1, valid_split=0.0, dataloader_kwargs={"num_workers": 0}, progress=False)
    tr.fit(num_epochs=2)
    assert tr._graphed is None


def test_fit_load_checkpoint_fit_recaptures(tmp_path):
    """A captured step holds raw pointers of the optimizer state it was captured with: load_checkpoint
    replaces the moment tensors, so the Trainer drops its graph there (and when fit() returns) and the next fit() captures
    anew - the continued run equals the eager continuation bit for bit."""
    import numpy as np
    import trainer as T

    class Cases(torch.utils.data.Dataset):
        def __init__(self, n=6):
            self.items = [{"image": O.synth_image((1, 1, 32, 32, 32), 900 + i)[0],
                           "label": O.phantom_labels(1, (32, 32, 32), 3)[0]} for i in range(n)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    def run(capture):
        torch.manual_seed(11)
        np.random.seed(11)
        ops._drop_counter[0] = 0                         # the Dropout3d draws are a function of (seed, counter)
        model = network.ResUnet3D(2, 8, 1, 3).to(DEV)
        tr = T.Trainer(model=model, optimizer=optim.Adam(model.parameters(), lr=1e-3), loss=L.HybirdLoss(),
                       dataset=Cases(), batch_size=1, valid_split=0.0, dataloader_kwargs={"num_workers": 0},
                       progress=False, capture_step=capture)
        tr.fit(num_epochs=1, save_dir=str(tmp_path / ("c%d" % int(bool(capture)))), use_amp=True, opt_level="bf16")
        assert tr._graphed is None                       # fit() leaves no graph (and no private pool) behind
        tr.load_checkpoint(str(tmp_path / ("c%d" % int(bool(capture)))) + "-last.pt")
        tr.fit(num_epochs=2, use_amp=True, opt_level="bf16")
        return {k: v.detach().float().cpu() for k, v in model.state_dict().items()}, tr.optimizer.state_dict()

    wa, sa = run(True)
    wb, sb = run(False)
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    for (ia, pa), (ib, pb) in zip(sorted(sa["state"].items()), sorted(sb["state"].items())):
        assert ia == ib and int(pa["step"]) == int(pb["step"]) and torch.equal(pa["exp_avg"].cpu(), pb["exp_avg"].cpu())
