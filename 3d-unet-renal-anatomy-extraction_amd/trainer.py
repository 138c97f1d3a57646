"""Drop-in `trainer` module: host-side driver of the native training step.

Keeps the reference `Trainer` surface (reference trainer.py:415-634: constructor arguments,
`fit(num_epochs, save_dir, use_amp, opt_level)`, `batch_loop`, `save_checkpoint` / `load_checkpoint`
with the same checkpoint keys, `get_lr` / `set_lr` / `summary`, `Subset`) and the pre-refactor
spelling the oldest script uses (reference run_train.py:71-96: `Trainer(criterion=, tr_transform=,
vd_transform=)`, `fit(dataset, batch_size=, epochs=, ...)`, `save` / `load`).

What changes underneath:
  * `use_amp=True` selects 16-bit storage / fp32 accumulation of the native kernels: fp16 with dynamic loss
    scaling for the reference's opt_level 'O1' (apex patched fp16 convs in, trainer.py:538-542, and scaled the
    loss, :492-493), or bf16 without loss scaling for opt_level='bf16'.
  * per-step `.item()` calls are deferred: the loss/metric scalars stay on the device and are read
    back once per `sync_every` steps (default: every step, the reference's behaviour), so the GPU
    is not drained between kernels.  The NaN rule is the reference's: a NaN step is excluded from
    the epoch mean but its optimizer step has already been applied (trainer.py:496 vs 505-506).
  * one process per GPU: when torch.distributed is initialised, gradients are averaged across
    ranks by `parallel.GradSync` (bucketed RCCL all-reduce on a side stream, overlapped with the
    rest of backward) and each rank draws its own samples.
  * apex / torchsummary / tensorboard / nibabel are optional.
"""
import gc
import math
import os

import numpy as np
import torch
from torch.optim import lr_scheduler

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None
try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:
    SummaryWriter = None

from loss import dice  # noqa: F401  (re-exported like the reference: trainer.dice)
import loss as _loss_mod
from inference import predict_per_patch, predict_case  # noqa: F401  (reference trainer.py:17, 101)


class _NullBar:
    def __init__(self, *a, **k):
        pass

    def reset(self, total=None):
        pass

    def set_description(self, *a, **k):
        pass

    def set_postfix(self, *a, **k):
        pass

    def update(self, *a, **k):
        pass

    def close(self):
        pass


class Subset(torch.utils.data.Subset):
    """Index subset that applies a transform to each fetched case."""

    def __init__(self, dataset, indices, transform):
        super().__init__(dataset, indices)
        self.transform = transform

    def __getitem__(self, idx):
        case = self.dataset[self.indices[idx]]
        return self.transform(case) if self.transform else case

    def __getitems__(self, indices):
        # torch >= 2.0 loaders fetch batches through __getitems__, and the inherited one reads self.dataset directly:
        # the transform of the reference's class (trainer.py:403-412, written for the per-item protocol) would be skipped
        return [self[i] for i in indices]


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() > 1


class _RankShardSampler(torch.utils.data.Sampler):
    """One process per GPU, one pass over the split per epoch: every epoch draws ONE permutation that is the same on all
    ranks (seed + epoch counter) and this rank takes every world-th case of it, wrapped so that all ranks take the same
    number of steps (the gradient exchange is collective).  The permutation changes with every __iter__, like the
    reference's `shuffle=True` loader (trainer.py:543-547)."""

    def __init__(self, n, rank, world, shuffle, seed, epoch=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, epoch

    def __len__(self):
        return -(-self.n // self.world)

    def __iter__(self):
        if self.shuffle:
            gen = torch.Generator()
            gen.manual_seed((self.seed + 104729 * self.epoch) % (1 << 63))
            order = torch.randperm(self.n, generator=gen).tolist()
        else:
            order = list(range(self.n))
        self.epoch += 1
        per_rank = len(self)
        order = (order * (-(-per_rank * self.world // self.n)))[:per_rank * self.world]
        return iter(order[self.rank::self.world])


class Trainer():
    def __init__(self, model, optimizer, loss=None, dataset=None, batch_size=10,
                 dataloader_kwargs={'num_workers': 2, 'pin_memory': True},
                 valid_split=0.2, num_samples=None, metrics=None, scheduler=None,
                 train_transform=None, valid_transform=None,
                 criterion=None, tr_transform=None, vd_transform=None, sync_every=1, progress=True,
                 capture_step=None):
        self.model = model
        self.optimizer = optimizer
        self.loss = loss if loss is not None else criterion
        if self.loss is None:
            raise TypeError("Trainer needs a loss (loss=... or the legacy criterion=...)")
        self.dataset = dataset
        self.metrics = metrics
        self.scheduler = scheduler
        self.train_transform = train_transform if train_transform is not None else tr_transform
        self.valid_transform = valid_transform if valid_transform is not None else vd_transform
        self.batch_size = batch_size
        self.dataloader_kwargs = {'batch_size': batch_size, **dataloader_kwargs}
        self.num_samples = num_samples
        self.valid_split = valid_split
        self.train_indices, self.valid_indices = [], []
        if dataset is not None:
            self._split_indices()
        self.device = next(model.parameters()).device
        self.best_result = {'loss': float('inf')}
        self.current_epoch = 0
        self.patience_counter = 0
        self.amp_state_dict = None
        self.use_amp = False
        self.save_dir = None
        self.num_epochs = 0
        self.sync_every = max(1, int(sync_every))
        self.progress = progress
        self.progress_bar = _NullBar()
        self._grad_sync = None
        self._scaler = None
        # capture_step: training batches of the usual shape are replayed from a hipGraph of the whole step
        # (graph.GraphedTrainStep - the same kernels on the same data, bit for bit; the host only copies the batch in and
        # one hipGraphLaunch replaces ~500 launches issued from Python).  Needs optim.Adam (its update kernel reads the
        # step's scalars from device memory), one process, bf16 / fp32 storage.  None (default): on whenever those hold -
        # with a caller-owned torch.optim.Adam, a loss that is not one of this package's fused losses (a user module may
        # have host-side effects a replay would skip), fp16 loss scaling or several ranks the loop is the eager one, and
        # it stays the fallback if a capture fails.  True: insist (raises when not possible).  False: never.
        self.capture_step = capture_step
        self._graphed = None
        self._capture_failed = False

    # ------------------------------------------------------------------ small helpers
    def _split_indices(self):
        # same draw as the reference: global numpy RNG, floor(valid_split * len) validation cases
        size = len(self.dataset)
        indices = list(range(size))
        split = int(np.floor(self.valid_split * size))
        np.random.shuffle(indices)
        self.train_indices = indices[split:]
        self.valid_indices = indices[:split]

    def get_lr(self, idx=0):
        return self.optimizer.param_groups[idx]['lr']

    def set_lr(self, lr, idx=0):
        self.optimizer.param_groups[idx]['lr'] = lr

    def summary(self, input_shape):
        try:
            from torchsummary import summary as _summary
            return _summary(self.model, input_shape)
        except ImportError:
            total = sum(p.numel() for p in self.model.parameters())
            trainable = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
            text = "%s: %d parameters (%d trainable), input %s" % (type(self.model).__name__, total, trainable,
                                                                   tuple(input_shape))
            print(text)
            return text

    # ------------------------------------------------------------------ the step
    def _flush(self, pending, results):
        """Read back deferred device scalars (one sync for the whole group)."""
        if not pending:
            return None
        keys = list(pending[0].keys())
        stacked = torch.stack([torch.stack([p[k].detach().float().reshape(()) for k in keys]) for p in pending])
        values = stacked.cpu().numpy()
        # the loss kernels' out-of-range label counts travelled in front of these scalars: F.one_hot's error
        # (reference loss.py:27) instead of a silently skipped NaN step, without a sync of its own
        _loss_mod.raise_on_bad_labels()
        last = None
        for row in values:
            result = {k: float(v) for k, v in zip(keys, row)}
            if not math.isnan(result['loss']):
                results.append(result)
            last = result
        pending.clear()
        return last

    def _graphed_step(self):
        if self.capture_step is False or self._capture_failed or self._grad_sync is not None:
            return None
        if self._graphed is None:
            import graph as graph_mod
            import optim as optim_mod
            if not isinstance(self.optimizer, optim_mod.Adam) or self.device.type != 'cuda':
                if self.capture_step:
                    raise TypeError("Trainer(capture_step=True) needs optim.Adam and a model on a HIP device")
                self._capture_failed = True       # auto mode: a caller-owned optimizer keeps the eager loop
                return None
            if self.capture_step is None and not isinstance(self.loss, _loss_mod._FusedLoss):
                # auto mode captures only what it knows to be free of host-side effects: a user-defined loss module (one
                # that logs, counts, branches on values) must keep running every step
                self._capture_failed = True
                return None
            self._graphed = graph_mod.GraphedTrainStep(self.model, self.loss, self.optimizer, scaler=self._scaler)
        return self._graphed

    def _release_graph(self):
        """Drop the captured step: its graph holds raw pointers of the optimizer state / gradients it was captured with
        and a private pool with a whole step's activations.  Called whenever those may change (load_checkpoint, a new
        fit() - possibly with another storage type or loss scaler) and when fit() returns."""
        if self._graphed is not None:
            stats = getattr(self, "graph_stats", None) or {"replays": 0, "eager_steps": 0}
            stats["replays"] += self._graphed.replays
            stats["eager_steps"] += self._graphed.eager_steps
            self.graph_stats = stats          # what the captured loop did, for logs and tests
            self._graphed.release()
            self._graphed = None

    def batch_loop(self, data_loader, is_train=True):
        results, pending = [], []
        self.progress_bar.reset(len(data_loader))
        self.progress_bar.set_description("Epoch %d/%d (LR %.2g)" % (self.current_epoch + 1, self.num_epochs,
                                                                     self.get_lr()))
        for batch_idx, batch in enumerate(data_loader):
            x = batch['image'].to(self.device, non_blocking=True)
            y = batch['label'].to(self.device, non_blocking=True)
            graphed = self._graphed_step() if is_train else None
            if graphed is not None:
                self.model.train()
                try:
                    loss = graphed(x, y)
                except Exception:
                    if self.capture_step or graphed.graph is not None:
                        raise
                    # auto mode and the capture itself failed (an op that cannot be captured): eager from here on
                    self._capture_failed = True
                    self._graphed = None
                    graphed = None
            if graphed is not None:
                y_pred = graphed.logits
                scalars = {'loss': loss.clone()}         # a static buffer: the next replay overwrites it
                if self.metrics is not None:
                    with torch.no_grad():
                        for key, metric_fn in self.metrics.items():
                            scalars[key] = metric_fn(y_pred, y)
                pending.append({k: (v if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in scalars.items()})
                if len(pending) >= self.sync_every:
                    last = self._flush(pending, results)
                    self.progress_bar.set_postfix(last)
                self.progress_bar.update()
                continue
            if is_train:
                self.model.train()
                y_pred = self.model(x)
            else:
                self.model.eval()
                with torch.no_grad():
                    y_pred = self.model(x)
            loss = self.loss(y_pred, y)
            if is_train:
                self.optimizer.zero_grad()
                if self._grad_sync is not None:
                    self._grad_sync.begin_step()
                if self._scaler is not None:
                    self._scaler.scale(loss).backward()      # reference: amp.scale_loss(loss, optimizer)
                else:
                    loss.backward()
                if self._grad_sync is not None:
                    self._grad_sync.finish_step()
                if self._scaler is not None:
                    self._scaler.step(self.optimizer)        # skipped on overflow, like apex's patched step
                else:
                    self.optimizer.step()
            scalars = {'loss': loss}
            if self.metrics is not None:
                with torch.no_grad():
                    for key, metric_fn in self.metrics.items():
                        scalars[key] = metric_fn(y_pred, y)
            pending.append({k: (v if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in scalars.items()})
            if len(pending) >= self.sync_every:
                last = self._flush(pending, results)
                self.progress_bar.set_postfix(last)
            self.progress_bar.update()
        self._flush(pending, results)

        mean_result = {}
        self._last_batches = len(results)
        if results:
            for key in results[0].keys():
                mean_result[key] = float(np.mean(np.array([r[key] for r in results])))
        else:
            mean_result = {'loss': float('nan')}
        name = 'train' if is_train else 'valid'
        if self.save_dir is not None and SummaryWriter is not None:
            writer = SummaryWriter(self.save_dir)
            for key, value in mean_result.items():
                writer.add_scalar('%s/%s' % (key, name), value, self.current_epoch)
            writer.close()
        return mean_result

    # ------------------------------------------------------------------ one process per GPU
    def _enter_distributed(self):
        """Make the ranks one job (the reference is single-process): rank 0's weights, optimizer hyper-parameters
        and train/valid split everywhere, gradient averaging on, per-rank sample streams.  Idempotent."""
        import torch.distributed as dist
        from parallel import make_grad_sync, broadcast_parameters
        if self._grad_sync is None:
            broadcast_parameters(self.model)
            # all ranks end up on one transport (RCCL when every rank can load it, torch.distributed otherwise)
            on_gpu = self.device.type == 'cuda' and dist.get_backend() != "gloo"
            self._grad_sync = make_grad_sync(self.model, transport=os.environ.get("RU3D_COMM", "rccl" if on_gpu else "torch"))
            # BatchNorm nets: each rank normalises its own sub-batch, as the reference's nn.DataParallel replicas do
            # (trainer.py:531-535; rank 0's running averages are the ones checkpointed); RU3D_SYNC_BN=1 pools the
            # statistics over the ranks instead (SyncBN, ops.set_bn_sync)
            if os.environ.get("RU3D_SYNC_BN", "0") == "1":
                import _ops as ops
                ops.set_bn_sync()
        box = [self.train_indices, self.valid_indices, self.current_epoch]
        dist.broadcast_object_list(box, src=0)
        self.train_indices, self.valid_indices, self.current_epoch = box

    def _epoch_mean_over_ranks(self, result):
        """Every rank must feed the scheduler and the best-checkpoint rule the same numbers: the mean over ALL batches of
        the epoch (sum and count are reduced, not a mean of per-rank means)."""
        if not _dist_ready() or not result:
            return result
        import torch.distributed as dist
        keys = sorted(result.keys())
        count = float(max(getattr(self, "_last_batches", 1), 0))
        vals = torch.tensor([[result[k] for k in keys]], dtype=torch.float64)
        vals = torch.nan_to_num(vals, nan=0.0)
        flags = torch.tensor([[0.0 if math.isnan(result[k]) else count for k in keys]], dtype=torch.float64)
        both = torch.cat([vals * flags, flags]).to(self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(both)
        both = both.cpu()
        return {k: (float(both[0, i] / both[1, i]) if both[1, i] > 0 else float('nan')) for i, k in enumerate(keys)}

    # ------------------------------------------------------------------ loaders
    def _loader(self, indices, transform, num_samples, shuffle):
        """One process per GPU: the ranks SHARE an epoch (the reference is one process, so an epoch is num_samples draws
        or one pass over the split, whatever the number of GPUs): rank r draws ceil(num_samples / world) samples from its
        own stream, or takes every world-th case of one common permutation (wrapped to equal length - every rank must
        take the same number of steps, the gradient exchange is collective)."""
        subset = Subset(self.dataset, indices, transform)
        kwargs = dict(self.dataloader_kwargs)
        if not _dist_ready():
            if num_samples is not None:
                sampler = torch.utils.data.RandomSampler(subset, True, num_samples)
                return torch.utils.data.DataLoader(subset, sampler=sampler, **kwargs)
            return torch.utils.data.DataLoader(subset, shuffle=shuffle, **kwargs)
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        base = torch.initial_seed() + 104729 * self.current_epoch
        if num_samples is not None:
            gen = torch.Generator()
            gen.manual_seed((base + 7919 * (rank + 1)) % (1 << 63))      # same split on every rank, different draws
            sampler = torch.utils.data.RandomSampler(subset, True, -(-num_samples // world), generator=gen)
            return torch.utils.data.DataLoader(subset, sampler=sampler, **kwargs)
        n = len(subset)
        if n == 0:
            return torch.utils.data.DataLoader(subset, shuffle=False, **kwargs)
        # a fresh common permutation per epoch (the loader is built once; the sampler counts its own epochs)
        sampler = _RankShardSampler(n, rank, world, shuffle, torch.initial_seed(), self.current_epoch)
        return torch.utils.data.DataLoader(subset, sampler=sampler, **kwargs)

    # ------------------------------------------------------------------ fit
    def fit(self, *args, num_epochs=None, save_dir=None, use_amp=False, opt_level='O1', **legacy):
        """fit(num_epochs=10, save_dir=None, use_amp=False, opt_level='O1')  - current API
        fit(dataset, batch_size=, epochs=, num_samples=, valid_split=, log_dir=, save_dir=, save_last=,
            save_best=, num_workers=, pin_memory=)                          - run_train.py spelling"""
        save_last = legacy.pop('save_last', True)
        save_best = legacy.pop('save_best', True)
        # keep the interpreter's full collection (it walks everything torch imported, ~90 ms) out of the step loop
        gc.collect()
        gc.freeze()
        if args and not isinstance(args[0], (int, np.integer)):
            self.dataset = args[0]
            args = args[1:]
            if 'valid_split' in legacy:
                self.valid_split = legacy.pop('valid_split')
            self._split_indices()
        if args:
            num_epochs = args[0]
            if len(args) > 1:
                save_dir = args[1]
        if 'epochs' in legacy:
            num_epochs = legacy.pop('epochs')
        if num_epochs is None:
            num_epochs = 10
        if 'batch_size' in legacy:
            self.batch_size = legacy.pop('batch_size')
            self.dataloader_kwargs['batch_size'] = self.batch_size
        for key in ('num_workers', 'pin_memory'):
            if key in legacy:
                self.dataloader_kwargs[key] = legacy.pop(key)
        if 'num_samples' in legacy:
            self.num_samples = legacy.pop('num_samples')
        log_dir = legacy.pop('log_dir', None)
        if save_dir is None and log_dir is not None:
            save_dir = log_dir
        if legacy:
            raise TypeError("fit() got unexpected arguments %s" % sorted(legacy))
        if self.dataset is None:
            raise ValueError("Trainer.fit: no dataset (pass dataset= to Trainer or as the first fit() argument)")

        self._release_graph()
        self.num_epochs = num_epochs
        self.use_amp = use_amp
        self.save_dir = save_dir
        if use_amp:
            # reference: amp.initialize(model, optimizer, opt_level=opt_level) - apex O1 patches fp16 convolutions in
            # and scales the loss dynamically.  Here: opt_level 'O1' / 'O2' / 'O3' select fp16 storage with fp32
            # accumulation in the HIP kernels plus optim.LossScaler (apex's schedule); opt_level 'bf16' (an extension,
            # also RU3D_AMP_DTYPE=bf16) selects bf16 storage, which needs no loss scaling.
            import network
            import optim as optim_mod
            level = str(opt_level).lower()
            if level == 'o0':
                amp_dtype = torch.float32
            elif level in ('bf16', 'bfloat16') or os.environ.get('RU3D_AMP_DTYPE', '').lower() in ('bf16', 'bfloat16'):
                amp_dtype = torch.bfloat16
            elif level in ('o1', 'o2', 'o3', 'fp16', 'float16'):
                amp_dtype = torch.float16
            else:
                raise ValueError("Trainer.fit: unknown opt_level %r" % (opt_level,))
            network.set_compute_dtype(self.model, amp_dtype)
            if amp_dtype == torch.float16:
                if self._scaler is None:
                    self._scaler = optim_mod.LossScaler()
                    if isinstance(self.amp_state_dict, dict):
                        self._scaler.load_state_dict(self.amp_state_dict)
            else:
                self._scaler = None
        if _dist_ready():
            self._enter_distributed()
        self.progress_bar = tqdm(total=0) if (tqdm is not None and self.progress) else _NullBar()

        train_loader = self._loader(self.train_indices, self.train_transform, self.num_samples, True)
        valid_loader = None
        if len(self.valid_indices) > 0:
            n_valid = round(self.num_samples * self.valid_split) if self.num_samples is not None else None
            valid_loader = self._loader(self.valid_indices, self.valid_transform, n_valid, False)

        for epoch in range(self.current_epoch, num_epochs):
            self.current_epoch = epoch
            result = self.batch_loop(train_loader, is_train=True)
            if valid_loader is not None:
                result = self.batch_loop(valid_loader, is_train=False)
            result = self._epoch_mean_over_ranks(result)
            if self.scheduler is not None:
                if isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau):
                    self.scheduler.step(result['loss'])
                else:
                    self.scheduler.step()
            if result['loss'] < self.best_result['loss'] - 1e-3:
                self.best_result = result
                if save_dir is not None and save_best:
                    self.save_checkpoint(save_dir + '-best.pt')
            if save_dir is not None and save_last:
                self.save_checkpoint(save_dir + '-last.pt')
        self.progress_bar.close()
        self._release_graph()
        return self.best_result

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, file_path):
        checkpoint = {'model_state_dict': self.model.state_dict(),
                      'optimizer_state_dict': self.optimizer.state_dict(),
                      'current_epoch': self.current_epoch,
                      'train_indices': self.train_indices,
                      'valid_indices': self.valid_indices,
                      'best_result': self.best_result}
        if self.scheduler is not None:
            checkpoint['scheduler_state_dict'] = self.scheduler.state_dict()
        if self.use_amp:
            # apex's loss-scaler state in the reference (trainer.py:617-618); bf16 needs none - the key is kept
            checkpoint['amp_state_dict'] = (self._scaler.state_dict() if self._scaler is not None
                                            else (self.amp_state_dict or {'ru3d': 'bf16'}))
        directory = os.path.dirname(file_path)
        if directory:
            os.makedirs(directory, exist_ok=True)
        if not _dist_ready() or torch.distributed.get_rank() == 0:
            torch.save(checkpoint, file_path)

    def load_checkpoint(self, file_path):
        try:
            checkpoint = torch.load(file_path, map_location=self.device, weights_only=False)
        except TypeError:  # older torch without weights_only
            checkpoint = torch.load(file_path, map_location=self.device)
        self._release_graph()       # optimizer.load_state_dict replaces the moment tensors a captured Adam points at
        self.model.load_state_dict(checkpoint['model_state_dict'])
        self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        self.current_epoch = checkpoint['current_epoch'] + 1
        self.train_indices = checkpoint['train_indices']
        self.valid_indices = checkpoint['valid_indices']
        self.best_result = checkpoint['best_result']
        if 'amp_state_dict' in checkpoint:
            self.amp_state_dict = checkpoint['amp_state_dict']
            if self._scaler is not None:
                self._scaler.load_state_dict(self.amp_state_dict)
        if 'scheduler_state_dict' in checkpoint and self.scheduler is not None:
            self.scheduler.load_state_dict(checkpoint['scheduler_state_dict'])

    # run_train.py spelling
    save = save_checkpoint
    load = load_checkpoint


# ---------------------------------------------------------------------------------------------------------------------
# Case-level drivers around predict_case (reference trainer.py:136-400): host glue, the arithmetic is in inference.py.
def batch_predict_case(load_dir, save_dir, model, target_spacing, normalize_stats, num_classes=3,
                       patch_size=(240, 240, 80), step_per_patch=4, data_range=None):
    """trainer.py:136-161"""
    from data import CaseDataset, save_pred
    cases = CaseDataset(load_dir)
    for i in (data_range if data_range is not None else range(len(cases))):
        case = predict_case(cases[i], model, target_spacing, normalize_stats, num_classes, patch_size,
                            step_per_patch, False)
        save_pred(case, save_dir)


def cascade_predict_case(case, coarse_model, coarse_target_spacing, coarse_normalize_stats, coarse_patch_size,
                         detail_model, detail_target_spacing, detail_normalize_stats, detail_patch_size,
                         num_classes=3, step_per_patch=4, region_threshold=10000, crop_padding=20, verbose=True):
    """trainer.py:164-245: a single-class coarse pass finds the regions of interest, the detail model predicts class
    probabilities inside each (padded) region, the regions' maps are averaged where they overlap and arg-maxed."""
    from data import regions_crop_case
    if verbose:
        print('Predicting the rough shape for further prediction...')
    case = predict_case(case, coarse_model, coarse_target_spacing, coarse_normalize_stats, 1, coarse_patch_size,
                        step_per_patch, verbose=verbose)
    regions = regions_crop_case(case, region_threshold, crop_padding, 'pred')
    num_classes = detail_model.out_channels
    orig_shape = case['image'].shape[:-1]
    total = np.zeros(list(orig_shape) + [num_classes])
    hits = np.zeros_like(total)
    if verbose:
        print('Cropping regions (%d)...' % len(regions))
    for idx, region in enumerate(regions):
        bbox, shape = region['bbox'], region['image'].shape[:-1]
        if verbose:
            print('Region {} {} predicting...'.format(idx, shape))
        region = predict_case(region, detail_model, detail_target_spacing, detail_normalize_stats, num_classes,
                              detail_patch_size, step_per_patch, verbose=verbose, one_hot=True)
        inside = tuple(slice(max(-bbox[d][0], 0), shape[d] - max(bbox[d][1] - orig_shape[d], 0)) for d in range(3))
        target = tuple(slice(max(bbox[d][0], 0), min(bbox[d][1], orig_shape[d])) for d in range(3))
        total[target] += region['pred'][inside]
        hits[target] += 1
    if verbose:
        print('Merging all regions...')
    seen = hits > 0
    total[seen] = total[seen] / hits[seen]
    if num_classes == 1:
        merged = np.around(np.squeeze(total, axis=-1))
    else:
        e = np.exp(total - total.max(axis=-1, keepdims=True))      # scipy.special.softmax, then argmax
        merged = np.argmax(e / e.sum(axis=-1, keepdims=True), axis=-1)
    case['pred'] = merged.astype(np.uint8)
    if verbose:
        print('All done!')
    return case


def cascade_predict(image_file, coarse_model, coarse_target_spacing, coarse_normalize_stats, coarse_patch_size,
                    detail_model, detail_target_spacing, detail_normalize_stats, detail_patch_size, air=-200, num_classes=3,
                    step_per_patch=4, region_threshold=10000, crop_padding=20, label_file=None, verbose=True):
    """trainer.py:248-302: load a NIfTI image, reorient + crop it to its non-air box, run the cascade on the crop, and
    put the mask back into a volume of the original file's grid.  As in the reference the final step applies the
    forward orientation once more (`apply_orientation(orig_pred, orient)`), which undoes the first one for the
    orientations that are their own inverse - every pure flip, and the usual axis swaps."""
    from data import apply_orientation, io_orientation, load_case, orient_crop_case
    orig_case = load_case(image_file, label_file)
    case = orient_crop_case(orig_case, air)
    case = cascade_predict_case(case, coarse_model, coarse_target_spacing, coarse_normalize_stats, coarse_patch_size,
                                detail_model, detail_target_spacing, detail_normalize_stats, detail_patch_size,
                                num_classes, step_per_patch, region_threshold, crop_padding, verbose)
    ornt = io_orientation(orig_case['affine'])
    order = ornt[:, 0].astype(int)
    orig_shape = np.take(np.array(orig_case['image'].shape[:3]), order)
    bbox = case['bbox']
    orig_pred = np.zeros(orig_shape, dtype=np.uint8)
    target = tuple(slice(max(bbox[d][0], 0), min(bbox[d][1], orig_shape[d])) for d in range(3))
    orig_pred[target] = case['pred']
    orig_case['pred'] = apply_orientation(orig_pred, ornt)
    if orig_case['image'].ndim == 3:
        orig_case['image'] = np.expand_dims(orig_case['image'], -1)
    return orig_case


def batch_cascade_predict(image_dir, save_dir, coarse_model, coarse_target_spacing, coarse_normalize_stats,
                          coarse_patch_size, detail_model, detail_target_spacing, detail_normalize_stats,
                          detail_patch_size, air=-200, num_classes=3, step_per_patch=4, region_threshold=10000,
                          crop_padding=20, data_range=None):
    """trainer.py:305-345: every file of `image_dir` through cascade_predict, masks written with save_pred."""
    from pathlib import Path
    from data import save_pred
    image_files = [path for path in sorted(Path(image_dir).iterdir()) if path.is_file()]
    for i in (data_range if data_range is not None else range(len(image_files))):
        case = cascade_predict(image_files[i], coarse_model, coarse_target_spacing, coarse_normalize_stats,
                               coarse_patch_size, detail_model, detail_target_spacing, detail_normalize_stats,
                               detail_patch_size, air, num_classes, step_per_patch, region_threshold, crop_padding,
                               None, False)
        save_pred(case, save_dir)


def evaluate_case(case):
    """trainer.py:348-356: Dice (loss.dice, alpha = beta = 0.5) of every foreground class of label vs pred."""
    out = []
    for c in range(int(case['label'].max())):
        p = np.array(case['pred'] == c + 1).astype(np.float32)
        g = np.array(case['label'] == c + 1).astype(np.float32)
        out.append(dice(torch.tensor(p), torch.tensor(g)).item())
    return out


def evaluate(label_file, pred_file):
    """trainer.py:359-368"""
    import nifti
    label, _, _ = nifti.load(label_file)
    pred, _, _ = nifti.load(pred_file)
    return evaluate_case({'label': label.astype(np.uint8), 'pred': pred.astype(np.uint8)})


def batch_evaluate(label_dir, pred_dir, data_range=None):
    """trainer.py:371-400"""
    from pathlib import Path
    label_files = sorted(Path(label_dir).glob('*.nii.gz'))
    pred_files = sorted(Path(pred_dir).glob('*.nii.gz'))
    results = [evaluate(label_files[i], pred_files[i])
               for i in (data_range if data_range is not None else range(len(label_files)))]
    print('\nThe mean dsc of each label:')
    for i, m in enumerate(np.array(results).mean(axis=0)):
        print("label_%d: %f" % (i + 1, m))
    return results
