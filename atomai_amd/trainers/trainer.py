"""BaseTrainer / SegTrainer on the MI355X hot path (reference: atomai/trainers/trainer.py:42-737).

Same attributes (net, optimizer, criterion, loss_acc, meta_state_dict, X_train ...), same batch schedule
(sklearn shuffle with batch_seed), same train/test-step order and checkpoint format.  Differences, all
deliberate: no zero-tensor TensorDataset, no nvidia-smi, Adam is the fused
flat optimizer (still a torch.optim.Adam), optional data-parallel gradient all-reduce over RCCL.
"""
import copy
import warnings
from collections import OrderedDict
from typing import Callable, List, Optional, Tuple, Type, Union

import os

import numpy as np
import torch

from .. import losses_metrics
from ..nets import init_fcnn_model
from ..optim import FusedAdam
from ..utils import (array2list, average_weights, gpu_usage_map, init_dataloaders, init_fcnn_dataloaders,
                     preprocess_training_image_data,
                     reset_bnorm, set_train_rng, weights_init)

warnings.filterwarnings("ignore", module="torch.nn.functional")


def _shuffle(arr: np.ndarray, random_state: int) -> np.ndarray:
    from sklearn.utils import shuffle           # defines the batch schedule (trainer.py:552-555)
    return shuffle(arr, random_state=random_state)


FUSE_LOSS = os.environ.get("AMX_FUSE_PX_LOSS", "1") != "0"       # host-side switch: the fused head + loss of train_step
EARLY_LOSS = os.environ.get("AMX_EARLY_LOSS", "1") != "0"       # host-side switch, read once at import (see _EarlyScalar)


class _EarlyScalar:
    """``loss.item()`` without draining the stream.  The reference reads the loss after ``optimizer.step()``
    (trainer.py:205-211); ``.item()`` there waits for everything queued so far — backward and Adam included — and only
    then can the host start preparing the next step, so the GPU idles for the host's per-step preamble (0.3-0.4 ms of
    a 19 ms U-Net step).  The value exists as soon as the forward pass is done: it is copied to pinned host memory on a
    side stream right then, and ``item()`` waits for that copy only.  Same value, same place in the program; the next
    step's kernels simply queue up behind the current backward.  AMX_EARLY_LOSS=0 restores the plain ``.item()``."""
    _pinned = {}                                   # device index -> [ring of pinned scalars, next slot]
    _RING = 64

    def __init__(self, t: torch.Tensor):
        self.t, self.ev = t, None
        if not t.is_cuda or not EARLY_LOSS:
            return
        from ..engine import aux_stream
        dev = t.device
        st = aux_stream(dev, 3)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        st.wait_event(ready)
        ring = _EarlyScalar._pinned.get(dev.index)
        if ring is None:
            ring = _EarlyScalar._pinned[dev.index] = [torch.empty(self._RING, dtype=torch.float32, pin_memory=True), 0]
        # every instance owns its slot: a second trainer (or a second loss of the same step) cannot overwrite a value
        # that has not been read yet (up to _RING values in flight per device)
        buf = ring[0][ring[1]:ring[1] + 1]
        ring[1] = (ring[1] + 1) % self._RING
        with torch.cuda.stream(st):
            buf.copy_(t.detach().reshape(1).float(), non_blocking=True)
        self.buf, self.ev = buf, torch.cuda.Event()
        self.ev.record(st)

    def item(self) -> float:
        if self.ev is None:
            return self.t.item()
        self.ev.synchronize()
        return float(self.buf[0])


class BaseTrainer:
    """Generic train loop: 1 cycle = 1 train mini-batch + 1 test mini-batch (trainer.py:233-251)."""

    def __init__(self):
        set_train_rng(1)
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.net = None
        self.criterion = None
        self.optimizer = None
        self.compute_accuracy = False
        self.full_epoch = True
        self.swa = False
        self.perturb_weights = False
        self.running_weights = {}
        self.training_cycles = 0
        self.batch_idx_train, self.batch_idx_test = [], []
        self.batch_size = 1
        self.nb_classes = None
        self.X_train, self.y_train = None, None
        self.X_test, self.y_test = None, None
        self.train_loader, self.test_loader = None, None
        self.data_is_set = False
        self.augdict = {}
        self.augment_fn = None
        self.filename = "model"
        self.print_loss = 1
        self.meta_state_dict = dict()
        self.loss_acc = {"train_loss": [], "test_loss": [], "train_accuracy": [], "test_accuracy": []}
        self.lr_scheduler = None
        self.accuracy_metrics = None
        self.plot_training_history = False
        self.dp = None                       # parallel.DataParallelGrads when world_size > 1

    # ------------------------------------------------------------------ small helpers
    def _reset_rng(self, seed: int) -> None:
        set_train_rng(seed)

    def _reset_weights(self) -> None:
        self.net.apply(weights_init)
        self.net.apply(reset_bnorm)

    def _reset_training_history(self) -> None:
        self.loss_acc = {"train_loss": [], "test_loss": [], "train_accuracy": [], "test_accuracy": []}

    def _delete_optimizer(self) -> None:
        self.optimizer = None

    def set_data(self, X_train, y_train, X_test, y_test, **kwargs) -> None:
        """Dataloaders (full_epoch) or lists of whole mini-batches to draw from (trainer.py:129-162)."""
        memory_alloc = kwargs.get("memory_alloc", 4)
        tor = lambda x: torch.from_numpy(x) if isinstance(x, np.ndarray) else x   # noqa: E731
        X_train, y_train, X_test, y_test = tor(X_train), tor(y_train), tor(X_test), tor(y_test)
        if self.full_epoch:
            self.train_loader, self.test_loader = init_dataloaders(
                X_train, y_train, X_test, y_test, self.batch_size, memory_alloc)
        else:
            self.X_train, self.y_train, self.X_test, self.y_test = array2list(
                X_train, y_train, X_test, y_test, self.batch_size, memory_alloc)
        self.data_is_set = True

    def set_model(self, model: Type[torch.nn.Module], nb_classes: int = None) -> None:
        self.net = model
        self.net.to(self.device)
        if self.nb_classes is None and nb_classes:           # as the reference (trainer.py:164-181): never wiped
            self.nb_classes = nb_classes

    def get_loss_fn(self, loss: Union[str, Callable] = 'mse', nb_classes: int = None):
        return losses_metrics.select_loss(loss, nb_classes)

    # ------------------------------------------------------------------ steps
    def train_step(self, feat: torch.Tensor, tar: torch.Tensor) -> Tuple[float]:
        """zero_grad -> forward -> loss -> backward -> [all-reduce] -> Adam (trainer.py:189-211)."""
        self.net.train()
        self.optimizer.zero_grad()
        feat, tar = feat.to(self.device), tar.to(self.device)
        prob = None
        ce = type(self.criterion) is losses_metrics.losses.CrossEntropyLoss and tar.dtype == torch.int64 and tar.ndim == 3
        bce = (type(self.criterion) is losses_metrics.losses.BCEWithLogitsLoss and tar.dtype == torch.float32
               and tar.ndim == 4 and tar.shape[1] == 1 and tar.shape[0] == feat.shape[0] and tar.shape[2:] == feat.shape[2:])
        if FUSE_LOSS and not self.compute_accuracy and (ce or bce) and hasattr(self.net, "forward_loss"):
            # head + loss + their backward in one pass over the last activation (nets/fcnn.py: forward_loss); the same values
            kind, out = self.net.forward_loss(feat, tar)
            loss = out if kind == "loss" else self.criterion(out, tar)
        else:
            prob = self.net(feat)
            loss = self.criterion(prob, tar)
        early = _EarlyScalar(loss)                 # the loss value starts its way to the host NOW (see the class)
        loss.backward()
        if self.dp is not None:
            self.dp.allreduce_grads()
        self.optimizer.step()
        if self.compute_accuracy:
            return (early.item(), self.accuracy_fn(tar, prob))
        return (early.item(),)

    def test_step(self, feat: torch.Tensor, tar: torch.Tensor) -> Tuple[float]:
        feat, tar = feat.to(self.device), tar.to(self.device)
        self.net.eval()
        with torch.no_grad():
            prob = self.net(feat)
            loss = self.criterion(prob, tar)
        if self.compute_accuracy:
            return (loss.item(), self.accuracy_fn(tar, prob))
        return (loss.item(),)

    def step(self, e: int) -> None:
        feat, tar = self.dataloader(self.batch_idx_train[e], mode='train')
        res = self.train_step(feat, tar)
        self.loss_acc["train_loss"].append(res[0])
        feat_, tar_ = self.dataloader(self.batch_idx_test[e], mode='test')
        res_ = self.test_step(feat_, tar_)
        self.loss_acc["test_loss"].append(res_[0])
        if self.compute_accuracy:
            self.loss_acc["train_accuracy"].append(res[1])
            self.loss_acc["test_accuracy"].append(res_[1])

    def step_full(self) -> None:
        """One pass over every mini-batch of both loaders (trainer.py:253-287)."""
        tot = {"tr": [0.0, 0.0, 0], "te": [0.0, 0.0, 0]}
        c = 0
        for feat, tar in self.train_loader:
            if self.augment_fn is not None:                  # every mini-batch, seeded by its index (trainer.py:262-266)
                feat, tar = self.augment_fn(feat, tar, seed=c)
            c += 1
            res = self.train_step(feat, tar)
            tot["tr"][0] += res[0]
            tot["tr"][1] += res[1] if self.compute_accuracy else 0
            tot["tr"][2] += 1
        c = 0
        for feat, tar in self.test_loader:
            if self.augment_fn is not None:                  # trainer.py:272-276
                feat, tar = self.augment_fn(feat, tar, seed=c)
            c += 1
            res = self.test_step(feat, tar)
            tot["te"][0] += res[0]
            tot["te"][1] += res[1] if self.compute_accuracy else 0
            tot["te"][2] += 1
        self.loss_acc["train_loss"].append(tot["tr"][0] / tot["tr"][2])
        self.loss_acc["test_loss"].append(tot["te"][0] / tot["te"][2])
        if self.compute_accuracy:
            self.loss_acc["train_accuracy"].append(tot["tr"][1] / tot["tr"][2])
            self.loss_acc["test_accuracy"].append(tot["te"][1] / tot["te"][2])

    def eval_model(self) -> None:
        self.net.eval()
        tot, acc, c = 0.0, 0.0, 0
        if self.full_epoch:
            batches = iter(self.test_loader)
        else:
            batches = (self.dataloader(i, mode='test') for i in range(len(self.X_test)))
        for feat, tar in batches:
            res = self.test_step(feat, tar)
            tot += res[0]
            acc += res[1] if self.compute_accuracy else 0
            c += 1
        print('Model (final state) evaluation loss:', np.around(tot / c, 4))
        if self.compute_accuracy:
            print('Model (final state) accuracy:', np.around(acc / c, 4))

    def dataloader(self, batch_num: int, mode: str = 'train') -> Tuple[torch.Tensor]:
        X, y = (self.X_test, self.y_test) if mode == 'test' else (self.X_train, self.y_train)
        feat, tar = X[batch_num][:self.batch_size], y[batch_num][:self.batch_size]
        if self.augment_fn is not None:
            feat, tar = self.augment_fn(feat, tar, seed=len(self.loss_acc["train_loss"]))
        return feat, tar

    # ------------------------------------------------------------------ checkpoint / stats
    def save_model(self, *args: str) -> None:
        """torch.save of {architecture kwargs, 'weights', 'optimizer'} (trainer.py:344-358)."""
        filename = args[0] if args else self.filename
        # always re-read: FusedAdam re-points the parameters into its flat buffer, so a state dict captured
        # at construction time (as the reference keeps, relying on aliasing) would be stale here
        self.meta_state_dict["weights"] = self.net.state_dict()
        if isinstance(self.optimizer, FusedAdam):    # stored as its torch.optim.Adam equivalent (portable)
            self.meta_state_dict["optimizer"] = self.optimizer.as_torch_adam()
        else:
            self.meta_state_dict["optimizer"] = self.meta_state_dict.get("optimizer", self.optimizer)
        if (self.dp is None or self.dp.rank == 0) and getattr(self, "_ens_rank", 0) == 0:
            torch.save(self.meta_state_dict, filename + '.tar')    # (sharded ensemble runs: rank 0's members only)

    def print_statistics(self, e: int, **kwargs) -> None:
        if self.dp is not None and self.dp.rank != 0:
            return
        name = self.accuracy_metrics or "Accuracy"
        mem = gpu_usage_map(torch.cuda.current_device()) if torch.cuda.is_available() else ['N/A ', ' N/A']
        msg = ['Epoch {}/{} ...'.format(e + 1, self.training_cycles),
               'Training loss: {} ...'.format(np.around(self.loss_acc["train_loss"][-1], 4)),
               'Test loss: {} ...'.format(np.around(self.loss_acc["test_loss"][-1], 4))]
        if self.compute_accuracy:
            msg += ['Train {}: {} ...'.format(name, np.around(self.loss_acc["train_accuracy"][-1], 4)),
                    'Test {}: {} ...'.format(name, np.around(self.loss_acc["test_accuracy"][-1], 4))]
        msg.append('GPU memory usage: {}/{}'.format(mem[0], mem[1]))
        print(*msg)

    def accuracy_fn(self, *args) -> None:
        raise NotImplementedError

    def weight_perturbation(self, e: int) -> None:
        a, gamma, e_p = (self.perturb_weights[k] for k in ("a", "gamma", "e_p"))
        if (e + 1) % e_p == 0:
            var = torch.tensor(a / (1 + e) ** gamma)
            for k, v in self.net.state_dict().items():
                v.copy_(v + v.new(v.shape).normal_(0, torch.sqrt(var)))
            if self.dp is not None:                  # per-rank noise: replicas follow rank 0's draw
                self.dp.broadcast_state(self.net)

    def save_running_weights(self, e: int) -> None:
        n_last = 5 if self.full_epoch else 30
        if self.training_cycles - e <= n_last:
            i_ = n_last - (self.training_cycles - e)
            self.running_weights[i_] = OrderedDict(
                # the tensors are views into FusedAdam's flat buffer: deepcopy would clone the WHOLE underlying
                # storage once per tensor; detach().cpu() copies only the view's elements
                (k, v.detach().cpu().clone()) for k, v in self.net.state_dict().items())

    def data_augmentation(self, augment_fn) -> None:
        self.augment_fn = augment_fn

    # ------------------------------------------------------------------ compile / run
    def compile_trainer(self, train_data=None, loss: Union[str, Callable] = 'ce',
                        optimizer: Optional[Type[torch.optim.Optimizer]] = None,
                        training_cycles: int = 1000, batch_size: int = 32,
                        compute_accuracy: bool = False, full_epoch: bool = False, swa: bool = False,
                        perturb_weights: bool = False, **kwargs):
        """Same arguments and defaults as the reference (trainer.py:441-565)."""
        self.full_epoch = full_epoch
        self.training_cycles = training_cycles
        self.batch_size = batch_size
        self.compute_accuracy = compute_accuracy
        self.swa = swa
        self.lr_scheduler = kwargs.get("lr_scheduler")
        alloc = kwargs.get("memory_alloc", 4)
        distributed = bool(kwargs.get("distributed", False))
        rank, world = 0, 1
        if distributed:
            # one process per GPU, rank r trains on ITS contiguous shard of X_train (equal
            # sizes: every step holds one collective), identical weights by broadcast, the reference's shuffle
            # schedule (trainer.py:552-555) seeded with batch_seed + rank, rank 0 saves.  Test data is not sharded:
            # every rank reports the same test loss as a single-process run.
            from .. import parallel
            rank, world, _ = parallel.init_distributed(force=True)
            if train_data is not None and world > 1:
                Xs, ys = parallel.shard_train_data(train_data[0], train_data[1], rank, world)
                train_data = (Xs, ys) + tuple(train_data[2:])
        if not self.data_is_set or kwargs.get("overwrite_train_data", True):
            self.set_data(*train_data, memory_alloc=alloc)
        self.perturb_weights = perturb_weights
        if self.perturb_weights:
            if self.meta_state_dict.get("batch_norm"):
                raise AssertionError("To use time-dependent weights perturbation, "
                                     "turn off the batch normalization layes")
            if isinstance(self.perturb_weights, bool):
                self.perturb_weights = {"a": .01, "gamma": 1.5, "e_p": 1 if self.full_epoch else 50}
        if self.optimizer is None:
            params = self.net.parameters()
            self.optimizer = FusedAdam(params, lr=1e-3) if optimizer is None else optimizer(params)
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer.prepare()
        if distributed and self.dp is None:
            if not isinstance(self.optimizer, FusedAdam):
                raise TypeError("distributed=True needs the flat-bucket optimizer (FusedAdam)")
            from ..parallel import DataParallelGrads, offset_rng_by_rank
            self.dp = DataParallelGrads(self.optimizer, self.net)
            offset_rng_by_rank(rank)
        if self.criterion is None:
            self.criterion = self.get_loss_fn(loss, self.nb_classes)
        if not self.full_epoch:
            seed = kwargs.get("batch_seed", 1)
            # the TRAINING schedule differs per rank (each rank walks its own shard); test data is not sharded, so its
            # schedule keeps the un-offset seed and every rank reports the test losses of a single-process run
            for name, data, sd in (("batch_idx_train", self.X_train, seed + rank), ("batch_idx_test", self.X_test, seed)):
                reps = self.training_cycles // len(data) + 1
                idx = np.arange(len(data)).repeat(reps)[:self.training_cycles]
                setattr(self, name, _shuffle(idx, sd))
        self.print_loss = kwargs.get("print_loss") or (1 if self.full_epoch else 100)
        self.accuracy_metrics = kwargs.get("accuracy_metrics")
        self.filename = kwargs.get("filename", "./model")
        self.plot_training_history = kwargs.get("plot_training_history", True)

    def select_lr(self, e: int) -> None:
        lr_i = self.lr_scheduler[e] if e < len(self.lr_scheduler) else self.lr_scheduler[-1]
        for g in self.optimizer.param_groups:
            g['lr'] = lr_i

    def run(self) -> Type[torch.nn.Module]:
        for e in range(self.training_cycles):
            if self.lr_scheduler is not None:
                self.select_lr(e)
            self.step_full() if self.full_epoch else self.step(e)
            if self.swa:
                self.save_running_weights(e)
            if self.perturb_weights:
                self.weight_perturbation(e)
            if e == 0 or (e + 1) % self.print_loss == 0 or e == self.training_cycles - 1:
                self.print_statistics(e)
        if not self.full_epoch:
            self.eval_model()
        if self.swa:
            print("Performing stochastic weight averaging...")
            self.net.load_state_dict(average_weights(self.running_weights))
            self.eval_model()
        self.save_model(self.filename + "_metadict_final")
        if self.plot_training_history:
            try:
                from ..utils.viz import plot_losses
                plot_losses(self.loss_acc["train_loss"], self.loss_acc["test_loss"])
            except Exception:        # plotting is optional tooling (matplotlib may be absent)
                pass
        return self.net

    def fit(self) -> None:
        _ = self.run()


class SegTrainer(BaseTrainer):
    """Trainer of the fully convolutional segmentation nets (trainer.py:610-737)."""

    def __init__(self, model: Union[Type[torch.nn.Module], str] = "Unet", nb_classes: int = 1,
                 **kwargs: Union[int, List, str, bool]) -> None:
        super().__init__()
        seed = kwargs.get("seed", 1)
        kwargs["batch_seed"] = kwargs.get("batch_seed", seed)
        self._batch_seed = kwargs["batch_seed"]
        set_train_rng(seed)
        self.nb_classes = nb_classes
        self.net, self.meta_state_dict = init_fcnn_model(model, self.nb_classes, **kwargs)
        self.net.to(self.device)
        if self.device == 'cpu':
            warnings.warn("No GPU found: the MI355X kernels cannot run (there is no CPU fallback)",
                          UserWarning)
        self.meta_state_dict["weights"] = self.net.state_dict()

    def set_data(self, X_train, y_train, X_test=None, y_test=None, **kwargs) -> None:
        if X_test is None or y_test is None:
            from sklearn.model_selection import train_test_split
            X_train, X_test, y_train, y_test = train_test_split(
                X_train, y_train, test_size=kwargs.get("test_size", .15), shuffle=True,
                random_state=kwargs.get("seed", 1))
        alloc = kwargs.get("memory_alloc", 4)
        if self.full_epoch:
            self.train_loader, self.test_loader, nb_classes = init_fcnn_dataloaders(
                X_train, y_train, X_test, y_test, self.batch_size, memory_alloc=alloc)
        else:
            (self.X_train, self.y_train, self.X_test, self.y_test,
             nb_classes) = preprocess_training_image_data(
                X_train, y_train, X_test, y_test, self.batch_size, alloc)
        self.data_is_set = True
        if self.nb_classes != nb_classes:
            raise AssertionError("Number of classes in initialized model is different from the number "
                                 "of classes contained in training data")

    def accuracy_fn(self, y, y_prob, *args):
        """Mean IoU of the mini-batch (trainer.py:727-737).  As in the reference the third positional argument of
        ``IoU`` — ``activation`` — receives ``self.nb_classes`` (truthy): the logits go through softmax / sigmoid."""
        return losses_metrics.IoU(y, y_prob, self.nb_classes).evaluate()
