"""CPU-only checks: the C-ABI library loads and exports every symbol include/ru3d.h declares (no
compute calls - there is no GPU here), the drop-in modules keep the reference's import surface,
state_dict contract and Trainer semantics, and the native path refuses CPU tensors loudly."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest
import torch

import _native as N
import loss as L
import network
import trainer as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "ru3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(ru3d_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert len(declared) >= 28
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    raw = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert N.lib.ru3d_version() == 201
    # pure host-side queries work without a device
    assert N.lib.ru3d_loss_state_bytes(3) > 0
    assert N.lib.ru3d_packed_weight_bytes(32, 32, 3, 1, N.ROLE_CONV_FWD, N.BF16) > 0
    assert N.lib.ru3d_packed_weight_bytes(32, 32, 5, 1, N.ROLE_CONV_FWD, N.BF16) == 0   # k=5 unsupported
    # argument validation happens before any launch
    rc = N.lib.ru3d_conv3d_fwd(None, None, None, None, None, 3, 1, N.F32, N.F32, None, 0, None)
    assert rc < 0 and b"conv3d_fwd" in N.lib.ru3d_last_error()


def test_ndhwc_descriptor_logic():
    t = N.new_act(2, 6, 3, 4, 5, torch.float32, "cpu")
    assert tuple(t.shape) == (2, 6, 3, 4, 5) and N.is_ndhwc(t)
    assert t.is_contiguous(memory_format=torch.channels_last_3d)
    sl = t[:, 2:4]
    assert N.is_ndhwc(sl)
    assert not N.is_ndhwc(torch.zeros(2, 6, 3, 4, 5))
    assert N.is_ndhwc(torch.zeros(2, 1, 3, 4, 5))              # C == 1: NCDHW is NDHWC
    assert not N.is_ndhwc(torch.zeros(2, 6, 3, 4))
    with pytest.raises(N.Ru3dError):
        N.desc(t)                                               # CPU tensor: refused
    assert N.dtype_code(torch.bfloat16) == N.BF16
    assert N.dtype_code(torch.float16) == N.F16                 # round 2: the reference's apex-O1 storage type
    with pytest.raises(N.Ru3dError):
        N.dtype_code(torch.float64)


def test_state_dict_contract_and_init_parity(golden_dir):
    info = json.load(open(os.path.join(golden_dir, "g5_checkpoint.json")))
    torch.manual_seed(0)
    model = network.ResUnet3D(num_pool=2, num_features=8, in_channels=1, out_channels=2)
    sd = model.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == info["model_state_dict"]
    assert list(sd.keys()) == list(info["model_state_dict"].keys())            # same order too
    # same construction order => same default initialisation as the reference under the same seed
    z = np.load(os.path.join(golden_dir, "g1_config1.npz"))
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), z["w/" + k]), k
    model.load_state_dict({k: torch.from_numpy(z["adam3/" + k]) for k in sd}, strict=True)
    m = network.ResUnet3D()
    assert (m.num_pool, m.num_features, m.in_channels, m.out_channels) == (4, 30, 1, 1)
    assert sum(p.numel() for p in network.ResUnet3D(4, 32, 1, 3).parameters()) == info["config2_num_parameters"]


def test_import_surface_matches_reference_scripts():
    for name in ("ResUnet3D", "ResAttrUnet3D", "ResAttrUnet3D2", "ResAttrBNUnet3D", "generate_paired_features",
                 "generate_paired_features2", "Unet", "ResBlock", "ResBlockStack", "ConvBlock", "ConvBlockStack",
                 "RecBlock", "ResRecBlock", "ConvTrans3D", "UpConcat", "AttBlock", "MaxPoolBlock", "none_fn"):
        assert hasattr(network, name), name
    for name in ("logits", "flatten_and_tranpose_C", "dice", "focal_loss", "Dice", "DiceLoss", "FocalLoss",
                 "HybirdLoss", "DiceCoef", "FocalDiceCoefLoss"):
        assert hasattr(L, name), name
    assert network.generate_paired_features(2, 8) == [[8, 8], [16, 16], [32, 32], [16, 16], [8, 8]]
    assert network.generate_paired_features2(2, 8) == [[8, 16], [16, 32], [32, 32], [16, 16], [8, 8]]
    # run_train.py constructs the net through the assembler with block classes as arguments
    net = network.Unet(in_channels=1, out_channels=3, paired_features=network.generate_paired_features(2, 4),
                       pool_block=network.ResBlock, pool_kwargs={'stride': 2}, up_kwargs={'attention': True},
                       encode_block=network.ResBlockStack, encode_kwargs_fn=lambda lv: {'num_stacks': max(lv, 1)},
                       decode_block=network.ResBlock)
    assert hasattr(net.up_blocks[0], "att_gate")
    # attention / BatchNorm variants exist and are wired as torch modules
    bn = network.ResAttrBNUnet3D(num_pool=1, num_features=4)
    assert isinstance(bn.net.pool_blocks[0].norm, torch.nn.BatchNorm3d) and not bn.net.pool_blocks[0]._native
    assert network.ResBlock(4, 4)._native and network.ConvTrans3D(8, 4)._native


def test_native_blocks_refuse_cpu_tensors():
    model = network.ResUnet3D(num_pool=1, num_features=4, in_channels=1, out_channels=2)
    with pytest.raises(N.Ru3dError, match="no CPU fallback"):
        model(torch.zeros(1, 1, 8, 8, 8))
    with pytest.raises(N.Ru3dError):
        network.ResBlock(4, 4)(torch.zeros(1, 4, 4, 4, 4))
    with pytest.raises(N.Ru3dError):
        L.HybirdLoss()(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 4, 4, 4, dtype=torch.long))
    # host-side functional dice (evaluate_case path) is plain host code
    p = torch.tensor([1.0, 0.0, 1.0, 1.0])
    g = torch.tensor([1, 0, 0, 1])
    assert abs(L.dice(p, g).item() - (2 + 1e-7) / (2 + 0.5 * 0 + 0.5 * 1 + 1e-7)) < 1e-6
    assert torch.allclose(L.logits(torch.zeros(1, 2, 3)), torch.full((1, 2, 3), 0.5))


def test_non_native_variants_run_on_cpu_as_torch_modules():
    blk = network.ConvBlockStack(2, 4, num_stacks=2)
    out = blk(torch.randn(1, 2, 6, 6, 6))
    assert tuple(out.shape) == (1, 4, 6, 6, 6)
    att = network.AttBlock(4)
    assert tuple(att(torch.randn(1, 4, 4, 4, 4), torch.randn(1, 4, 4, 4, 4)).shape) == (1, 4, 4, 4, 4)
    rr = network.ResRecBlock(2, 4, dropout_op=None)
    assert tuple(rr(torch.randn(1, 2, 4, 4, 4)).shape) == (1, 4, 4, 4, 4)


# --------------------------------------------------------------------------- Trainer plumbing (toy torch model on CPU)
class _ToyData(torch.utils.data.Dataset):
    def __init__(self, n=8, nan_at=None):
        g = torch.Generator().manual_seed(0)
        self.x = torch.randn(n, 1, 4, 4, 4, generator=g)
        self.y = (self.x[:, 0] > 0).long()
        self.nan_at = nan_at

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        x = self.x[i].clone()
        if self.nan_at == i:
            x[:] = float("nan")
        return {"image": x, "label": self.y[i], "case_id": "c%d" % i, "affine": np.eye(4)}


class _ToyLoss(torch.nn.Module):
    def forward(self, pred, target):
        return torch.nn.functional.cross_entropy(pred, target)


def _toy_trainer(tmp_path, **kw):
    model = torch.nn.Conv3d(1, 2, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.2, patience=25)
    np.random.seed(0)
    tr = T.Trainer(model=model, optimizer=opt, loss=_ToyLoss(), dataset=kw.pop("dataset", _ToyData()), batch_size=2,
                   dataloader_kwargs={"num_workers": 0}, scheduler=sch, progress=False,
                   metrics={"acc": lambda p, t: (p.argmax(1) == t).float().mean()}, **kw)
    return tr


def test_trainer_fit_checkpoint_resume(tmp_path):
    tr = _toy_trainer(tmp_path, valid_split=0.25, num_samples=6)
    assert len(tr.valid_indices) == 2 and len(tr.train_indices) == 6
    save = str(tmp_path / "run")
    best = tr.fit(num_epochs=2, save_dir=save)
    assert set(best) == {"loss", "acc"} and math.isfinite(best["loss"])
    ckpt = torch.load(save + "-last.pt", weights_only=False)
    assert set(ckpt) == {"model_state_dict", "optimizer_state_dict", "current_epoch", "train_indices",
                         "valid_indices", "best_result", "scheduler_state_dict"}
    assert ckpt["current_epoch"] == 1 and os.path.exists(save + "-best.pt")
    tr2 = _toy_trainer(tmp_path, valid_split=0.25, num_samples=6)
    tr2.load_checkpoint(save + "-last.pt")
    assert tr2.current_epoch == 2 and tr2.train_indices == ckpt["train_indices"]
    tr2.fit(num_epochs=3)
    assert tr2.current_epoch == 2          # ran exactly epoch index 2
    assert tr.get_lr() == 1e-2
    tr.set_lr(5e-3)
    assert tr.get_lr() == 5e-3
    assert "parameters" in tr.summary((1, 4, 4, 4))


def test_subset_applies_its_transform_through_a_dataloader():
    """reference trainer.py:403-412: Subset.__getitem__ applies the transform.  torch >= 2.0 DataLoaders fetch through
    __getitems__, which torch.utils.data.Subset implements by reading the wrapped dataset directly - the drop-in class
    must keep the transform on that route too (also when nested in another Subset, as the rank shards are)."""
    class Ids(torch.utils.data.Dataset):
        def __len__(self):
            return 6

        def __getitem__(self, i):
            return {"id": i}
    seen = []
    sub = T.Subset(Ids(), [5, 3, 1, 0], lambda case: (seen.append(case["id"]), {"id": case["id"] * 10})[1])
    got = [b["id"].tolist() for b in torch.utils.data.DataLoader(sub, batch_size=2)]
    assert got == [[50, 30], [10, 0]] and seen == [5, 3, 1, 0]
    del seen[:]
    nested = torch.utils.data.Subset(sub, [3, 0])
    got = [b["id"].tolist() for b in torch.utils.data.DataLoader(nested, batch_size=2)]
    assert got == [[0, 50]] and seen == [0, 5]


def test_loss_scaler_state_is_apex_layout_in_both_directions():
    """The reference checkpoints apex's amp.state_dict() as 'amp_state_dict' (trainer.py:617-618):
    {'loss_scaler0': {'loss_scale', 'unskipped'}}.  LossScaler writes that layout, reads it (a reference checkpoint),
    still reads its own round-2 flat layout, and caps the scale at apex's max_loss_scale = 2**24."""
    import optim
    a = optim.LossScaler()
    a.loss_scale, a.growth_tracker, a.skipped_steps = 1024.0, 17, 3
    st = a.state_dict()
    assert st["loss_scaler0"] == {"loss_scale": 1024.0, "unskipped": 17}
    assert [k for k in st if k.startswith("loss_scaler")] == ["loss_scaler0"]       # what apex's loader iterates over
    b = optim.LossScaler()
    b.load_state_dict(st)
    assert (b.loss_scale, b.growth_tracker, b.skipped_steps) == (1024.0, 17, 3)
    c = optim.LossScaler()
    c.load_state_dict({"loss_scaler0": {"loss_scale": 32768.0, "unskipped": 1999}})   # written by the reference
    assert (c.loss_scale, c.growth_tracker, c.skipped_steps) == (32768.0, 1999, 0)
    c.load_state_dict({"loss_scale": 512.0, "growth_tracker": 5, "skipped_steps": 2, "ru3d": "fp16"})   # round 2
    assert (c.loss_scale, c.growth_tracker, c.skipped_steps) == (512.0, 5, 2)
    c.load_state_dict({"loss_scaler0": {"loss_scale": 2.0 ** 30, "unskipped": 0}})
    assert c.loss_scale == 2.0 ** 24
    c.load_state_dict({"ru3d": "bf16"})                                             # a bf16 run's placeholder: ignored
    assert c.loss_scale == 2.0 ** 24


def test_trainer_nan_step_is_skipped_but_applied(tmp_path):
    # reference trainer.py:496 vs 505-506: the optimizer step happens before the NaN test
    tr = _toy_trainer(tmp_path, dataset=_ToyData(4, nan_at=1), valid_split=0.0)
    tr.dataloader_kwargs["batch_size"] = 1
    tr.num_epochs = 1
    loader = tr._loader(list(range(4)), None, None, False)
    res = tr.batch_loop(loader, is_train=True)
    assert math.isfinite(res["loss"])          # NaN step excluded from the mean
    assert any(torch.isnan(p).any() for p in tr.model.parameters())   # ... but its update was applied


def test_trainer_legacy_spelling(tmp_path):
    model = torch.nn.Conv3d(1, 2, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    tr = T.Trainer(model=model, optimizer=opt, criterion=_ToyLoss(), tr_transform=None, vd_transform=None,
                   progress=False)
    assert tr.loss is not None and tr.dataset is None
    log = str(tmp_path / "legacy")
    tr.fit(_ToyData(), batch_size=2, epochs=1, num_samples=4, log_dir=log, save_dir=log, save_last=True,
           save_best=False, num_workers=0, pin_memory=False)
    assert os.path.exists(log + "-last.pt") and not os.path.exists(log + "-best.pt")
    tr.save(str(tmp_path / "init.pt"))
    tr.load(str(tmp_path / "init.pt"))
    with pytest.raises(TypeError):
        T.Trainer(model=model, optimizer=opt)


def test_deferred_sync_gives_same_means(tmp_path):
    torch.manual_seed(0)
    a = _toy_trainer(tmp_path, valid_split=0.0)
    torch.manual_seed(0)
    b = _toy_trainer(tmp_path, valid_split=0.0, sync_every=4)
    b.model.load_state_dict(a.model.state_dict())
    for t in (a, b):
        t.num_epochs = 1
    la = a._loader(list(range(8)), None, None, False)
    lb = b._loader(list(range(8)), None, None, False)
    ra, rb = a.batch_loop(la), b.batch_loop(lb)
    assert ra.keys() == rb.keys() and abs(ra["loss"] - rb["loss"]) < 1e-6


def test_sliding_window_plan_matches_numpy_integer_arange():
    """inference.window_centres restates np.arange(start, end + 1e-8, step, dtype=int) (reference
    trainer.py:38-40, np.int == int): same centres for every length/patch/step combination tried, every
    window inside the padded volume."""
    import inference as I
    from oracle import unet_oracle as O
    for patch in (8, 16, 80, 96, 128, 240):
        for spp in (1, 2, 3, 4, 5):
            for length in list(range(patch, patch + 40)) + [2 * patch, 3 * patch + 1, 512, 517]:
                if length < patch:
                    continue
                start, end = patch // 2, length - patch // 2
                num_steps = np.ceil((end - start) / (patch / spp))
                step = np.array([(end - start) / (num_steps + 1e-8)])
                step[step == 0] = 9999999
                want = np.arange(np.int64(start), np.int64(end) + 1e-8, step[0], dtype=int).tolist()
                got = I.window_centres(length, patch, spp)
                assert got == want, (length, patch, spp, got, want)
                assert got[0] - patch // 2 == 0 and got[-1] + patch // 2 <= length
    # pad / crop offsets against crop_pad's box arithmetic, incl. the odd-difference one-voxel shift
    for orig, patch in ((11, 16), (12, 16), (16, 16), (21, 16), (1, 96)):
        full = I.padded_shape((orig,) * 3, (patch,) * 3)[0]
        ramp = np.arange(1, orig + 1, dtype=np.float32)
        padded = O.center_crop_pad(ramp, [full])
        lo = I.pad_offset((orig,) * 3, (full,) * 3)[0]
        assert padded.shape == (full,) and padded[lo] == 1 and (lo == 0 or padded[lo - 1] == 0)
        back = O.center_crop_pad(padded, [orig])
        co = I.crop_offset((orig,) * 3, (full,) * 3)[0]
        assert np.array_equal(back, padded[co:co + orig])
    assert T.predict_per_patch is I.predict_per_patch


def test_predict_per_patch_refuses_cpu_model_and_odd_patch():
    import inference as I
    model = network.ResUnet3D(1, 4, 1, 2)
    with pytest.raises(N.Ru3dError):
        I.predict_per_patch(np.zeros((8, 8, 8, 1), np.float32), model, 2, (8, 8, 8), 2, False)


# --------------------------------------------------------------------------- round 2: launcher / device plumbing
def test_bench_refuses_gpus_that_disagree_with_the_launcher():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"WORLD_SIZE=3" in p.stderr


def test_bench_algorithmic_work_matches_survey_table():
    """ALG_FLOPS / ALG_BYTES per step for the four configurations."""
    import bench
    for cfg, flops, nbytes in (((2, 8, 2, 1, (32,) * 3, 4), 3.49e9, 75e6),
                               ((4, 32, 3, 2, (128,) * 3, 2), 9.004e12, 19.98e9),
                               ((4, 30, 3, 2, (160, 160, 80), 2), 7.730e12, 18.28e9),
                               ((5, 64, 3, 1, (192,) * 3, 2), 63.36e12, 76.53e9)):
        f, b = bench.alg_work(*cfg)
        assert abs(f - flops) / flops < 2e-3 and abs(b - nbytes) / nbytes < 3e-3, (cfg, f, b)


def test_stream_follows_the_operands_device(monkeypatch):
    """the stream handed to the C ABI is the current stream of the operands' device, not of the
    thread's current device."""
    import types
    seen = []

    def fake_current_stream(device=None):
        seen.append(device)
        return types.SimpleNamespace(cuda_stream=1234)

    monkeypatch.setattr(torch.cuda, "current_stream", fake_current_stream)
    N.note_device(torch.device("cuda", 1))
    assert N.stream().value == 1234 and seen[-1] == torch.device("cuda", 1)
    N.note_device(torch.device("cuda", 0))
    N.stream()
    assert seen[-1] == torch.device("cuda", 0)


def test_adam_capture_scalars_follow_step_counts_and_lr():
    """optim.Adam's captured-step interface (graph.GraphedTrainStep): the scalars uploaded before replay k are those of
    eager step k - bias corrections from the advancing step count, the current lr, sqrt(bias_corr2) as the eager launch
    takes it - and the counts flow back into state_dict()."""
    import math
    import optim
    p = torch.nn.Parameter(torch.zeros(4))
    opt = optim.Adam([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    with pytest.raises(ValueError):
        opt.begin_capture(torch.zeros(2, 8))                 # one row per param group
    opt._captured = {"hyper": None, "steps": {0: 3.0}}       # as the capture of step 4 leaves it
    opt.state[p] = {"step": torch.tensor(4.0), "exp_avg": torch.zeros(4), "exp_avg_sq": torch.zeros(4)}
    row = torch.zeros(1, 8)
    opt.replay_scalars(row)                                  # first replay = step 4
    assert opt._captured["steps"][0] == 4.0
    exp = [1e-3, 0.9, 0.999, 1e-8, 1 - 0.9 ** 4, 1 - 0.999 ** 4, 1.0]
    assert torch.allclose(row[0, :7], torch.tensor(exp, dtype=torch.float32), rtol=1e-6, atol=0)
    assert abs(float(row[0, 7]) - math.sqrt(float(row[0, 5]))) < 1e-7
    opt.param_groups[0]["lr"] = 5e-4                         # a scheduler between replays
    opt.replay_scalars(row, grad_scale=0.5)
    assert float(row[0, 0]) == pytest.approx(5e-4) and float(row[0, 6]) == 0.5
    assert float(row[0, 4]) == pytest.approx(1 - 0.9 ** 5, rel=1e-6)
    assert float(opt.state_dict()["state"][0]["step"]) == 5.0


def test_graphed_step_needs_the_fused_optimizer():
    import graph
    m = torch.nn.Linear(2, 2)
    with pytest.raises(TypeError):
        graph.GraphedTrainStep(m, torch.nn.MSELoss(), torch.optim.Adam(m.parameters()))


def test_rank_shard_sampler_reshuffles_every_epoch_and_covers_the_split():
    """Multi-rank training without num_samples: the loader is built once, so the sampler itself must draw a new
    common permutation per epoch (the reference's shuffle=True loader, trainer.py:543-547); every epoch the ranks' shards
    together cover the split, with equal step counts."""
    import trainer as T
    n, world = 11, 3
    samplers = [T._RankShardSampler(n, r, world, True, seed=1234, epoch=0) for r in range(world)]
    epochs = []
    for _ in range(3):
        shards = [list(s) for s in samplers]
        assert len({len(sh) for sh in shards}) == 1 and len(shards[0]) == len(samplers[0]) == 4
        assert set(i for sh in shards for i in sh) == set(range(n))      # wrapped to 12 draws: one case twice
        epochs.append(shards)
    assert epochs[0][0] != epochs[1][0] and epochs[1][0] != epochs[2][0]   # a rank sees a new order every epoch
    # a resumed run (epoch counter from the checkpoint) continues the sequence
    resumed = T._RankShardSampler(n, 0, world, True, seed=1234, epoch=2)
    assert list(resumed) == epochs[2][0]
    # shuffle=False (validation): the same contiguous deal every epoch
    v = T._RankShardSampler(n, 1, world, False, seed=0)
    assert list(v) == list(v) == [1, 4, 7, 10]
