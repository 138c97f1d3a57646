"""`Trainer.fit` / `batch_loop` (reference trainer.py:465-604) driving the HIP model on a real MI355X: the loss
trajectory equals the CPU oracle's train_step + adam_step on the same samples (fp32 parity mode, Dropout3d p = 0),
`use_amp=True` switches the kernels to bf16 storage, deferred scalar read-back (`sync_every`) does not change the
numbers, and a checkpoint resumes to a bit-identical next step.  Needs a real MI355X: run with `-m gpu`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import loss as L  # noqa: E402
import network  # noqa: E402
import optim  # noqa: E402
import trainer as T  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


class Cases(torch.utils.data.Dataset):
    """In-memory synthetic cases in the reference's sample format (data.py: dict with 'image' [C,W,H,D], 'label')."""

    def __init__(self, n=3):
        self.items = []
        self.log = []
        for i in range(n):
            x = O.synth_image((1, 1, 32, 32, 32), 500 + i)[0]
            y = O.phantom_labels(1, (32, 32, 32), 2)[0]
            if i % 2:
                y = y.flip(0)
            self.items.append({"image": x, "label": y})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        self.log.append(int(i))
        return self.items[i]


class Recorder(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.values = []

    def forward(self, p, t):
        v = self.inner(p, t)
        self.values.append(v.detach())
        return v


def _model(seed=0):
    torch.manual_seed(seed)
    m = network.ResUnet3D(2, 8, 1, 2).to(DEV)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout3d):
            mod.p = 0.0            # train mode without the random masks: the oracle can replay the steps
    return m


def _fit(sync_every, use_amp, epochs=2, fused=False, save_dir=None):
    model = _model()
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = (optim.Adam if fused else torch.optim.Adam)(model.parameters(), lr=1e-4)
    data = Cases(3)
    rec = Recorder(L.HybirdLoss())
    torch.manual_seed(11)
    np.random.seed(11)
    tr = T.Trainer(model=model, optimizer=opt, loss=rec, dataset=data, batch_size=1, valid_split=0.0,
                   dataloader_kwargs={"num_workers": 0}, metrics={"dice": L.Dice()}, sync_every=sync_every,
                   progress=False)
    best = tr.fit(num_epochs=epochs, save_dir=save_dir, use_amp=use_amp)
    torch.cuda.synchronize()
    return tr, model, w0, data, [float(v) for v in rec.values], best


def _oracle_losses(w0, data, order):
    w = {k: v.clone() for k, v in w0.items()}
    state = {}
    out = []
    for i in order:
        it = data.items[i]
        loss, _, grads = O.train_step(w, it["image"][None], it["label"][None], 2)
        O.adam_step(w, grads, state)
        out.append(float(loss))
    return out, w


def test_fit_fp32_matches_oracle_trajectory():
    tr, model, w0, data, losses, best = _fit(sync_every=1, use_amp=False)
    assert len(losses) == 6 and len(data.log) == 6          # 2 epochs x 3 steps
    ref, w_ref = _oracle_losses(w0, data, data.log)
    assert np.allclose(losses, ref, rtol=0, atol=3e-5), (losses, ref)
    assert abs(best["loss"] - np.mean(ref[3:])) < 3e-5 or abs(best["loss"] - np.mean(ref[:3])) < 3e-5
    assert "dice" in best and 0.0 <= best["dice"] <= 1.0
    assert model.net.compute_dtype == torch.float32
    # parameters after six Adam steps: bounded as in test_g1_adam_three_steps
    for k, p in model.state_dict().items():
        assert float((p.cpu() - w_ref[k]).abs().max()) <= 2.1e-4 * 6, k


def test_fit_deferred_readback_and_fused_adam_same_numbers():
    _, _, _, data1, losses1, best1 = _fit(sync_every=1, use_amp=False)
    _, _, _, data3, losses3, best3 = _fit(sync_every=3, use_amp=False)
    assert data1.log == data3.log
    assert losses1 == losses3 and best1 == best3            # same kernels, same order: same bits
    _, _, _, dataf, lossesf, _ = _fit(sync_every=3, use_amp=False, fused=True)
    assert dataf.log == data1.log
    assert np.allclose(lossesf, losses1, rtol=0, atol=2e-5)


def _fit_amp(opt_level, fused):
    model = _model()
    w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = (optim.Adam if fused else torch.optim.Adam)(model.parameters(), lr=1e-4)
    data = Cases(3)
    rec = Recorder(L.HybirdLoss())
    torch.manual_seed(11)
    np.random.seed(11)
    tr = T.Trainer(model=model, optimizer=opt, loss=rec, dataset=data, batch_size=1, valid_split=0.0,
                   dataloader_kwargs={"num_workers": 0}, progress=False)
    tr.fit(num_epochs=2, use_amp=True, opt_level=opt_level)
    torch.cuda.synchronize()
    return tr, model, w0, data, [float(v) for v in rec.values]


@pytest.mark.parametrize("opt_level,dtype,fused", [("O1", torch.float16, True), ("O1", torch.float16, False),
                                                   ("bf16", torch.bfloat16, True)])
def test_fit_use_amp_maps_opt_level_to_16_bit_storage(opt_level, dtype, fused):
    """use_amp=True, opt_level='O1' (the reference's apex call, trainer.py:538-542) -> fp16 storage + dynamic loss
    scaling; opt_level='bf16' -> bf16 storage, no scaler.  Either way the loss (fp32, from fp32 logits) follows the
    fp32 oracle within the 16-bit storage noise and the weights move."""
    tr, model, w0, data, losses = _fit_amp(opt_level, fused)
    assert model.net.compute_dtype == dtype
    assert (tr._scaler is not None) == (dtype == torch.float16)
    ref, _ = _oracle_losses(w0, data, data.log)
    assert all(np.isfinite(losses))
    tol = 2e-2 if dtype == torch.bfloat16 else 5e-3
    assert np.allclose(losses, ref, rtol=0, atol=tol), (losses, ref)
    assert any(abs(a - b) > 1e-7 for a, b in zip(losses, ref))      # and it is not the fp32 path
    moved = sum(int(not torch.equal(v.cpu(), w0[k])) for k, v in model.state_dict().items())
    if dtype == torch.float16:
        # apex semantics: steps whose gradients overflowed at the current scale were skipped, the scale halved
        sc = tr._scaler
        assert sc.loss_scale == 65536.0 * 0.5 ** sc.skipped_steps and sc.skipped_steps < 6
        assert sc.growth_tracker == 6 - sc.skipped_steps
    assert moved > 20


def test_checkpoint_resume_identical_next_step(tmp_path):
    save = os.path.join(str(tmp_path), "run")
    tr, model, _, data, _, _ = _fit(sync_every=1, use_amp=False, epochs=1, save_dir=save)
    assert os.path.exists(save + "-last.pt")
    model2 = _model(seed=5)                               # different weights: the checkpoint must overwrite them
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-4)
    tr2 = T.Trainer(model=model2, optimizer=opt2, loss=L.HybirdLoss(), dataset=Cases(3), batch_size=1,
                    valid_split=0.0, dataloader_kwargs={"num_workers": 0}, progress=False)
    tr2.load_checkpoint(save + "-last.pt")
    assert tr2.current_epoch == 1 and tr2.train_indices == tr.train_indices
    batch = {"image": data.items[0]["image"][None], "label": data.items[0]["label"][None]}
    r1 = tr.batch_loop([batch], is_train=True)
    r2 = tr2.batch_loop([batch], is_train=True)
    torch.cuda.synchronize()
    assert r1["loss"] == r2["loss"]
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k


def test_fit_raises_on_out_of_range_labels():
    """A label >= num_classes does not vanish into a skipped NaN step: Trainer raises F.one_hot's error (reference
    loss.py:27 through trainer.py:486) at its read-back of that step's scalars."""
    model = _model()
    data = Cases(2)
    data.items[1]["label"] = data.items[1]["label"].clone()
    data.items[1]["label"][0, 0, 0] = 2          # the model has two classes
    L.raise_on_bad_labels(wait=True)
    tr = T.Trainer(model=model, optimizer=torch.optim.Adam(model.parameters(), lr=1e-4), loss=L.HybirdLoss(),
                   dataset=data, batch_size=1, valid_split=0.0, dataloader_kwargs={"num_workers": 0}, sync_every=1,
                   progress=False)
    with pytest.raises(RuntimeError, match="Class values must be smaller than num_classes"):
        tr.fit(num_epochs=1)
    L.raise_on_bad_labels(wait=True)


def test_fit_raises_on_out_of_range_labels_under_graph_replay():
    """The same guarantee on the DEFAULT training path: with optim.Adam the step is captured and replayed, the
    loss kernel's bad-label count is read back behind every replay and Trainer raises at its next read-back."""
    class LateBad(Cases):
        calls = 0

        def __getitem__(self, i):
            item = dict(super().__getitem__(i))
            LateBad.calls += 1
            if LateBad.calls >= 4:                   # steps 1-2 are the eager warm-up, step 3 the capture, 4.. replays
                item["label"] = item["label"].clone()
                item["label"][0, 0, 0] = 2           # the model has two classes
            return item

    model = _model()
    L.raise_on_bad_labels(wait=True)
    tr = T.Trainer(model=model, optimizer=optim.Adam(model.parameters(), lr=1e-4), loss=L.HybirdLoss(),
                   dataset=LateBad(6), batch_size=1, valid_split=0.0, dataloader_kwargs={"num_workers": 0},
                   sync_every=1, progress=False)
    with pytest.raises(RuntimeError, match="Class values must be smaller than num_classes"):
        tr.fit(num_epochs=1)
    assert tr._graphed is not None and tr._graphed.replays >= 2      # the step that raised was a replay
    L.raise_on_bad_labels(wait=True)
    tr._release_graph()
