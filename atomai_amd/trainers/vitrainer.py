"""viBaseTrainer: the VAE train loop (reference: atomai/trainers/vitrainer.py:19-396).

Same attributes (encoder_net, decoder_net, optim, loss_history, metadict, train_iterator ...), same
DataLoader semantics (shuffle, drop_last, device-resident data), same checkpoint format.  Adam is the fused
flat optimizer (decoder parameters first, as in the reference) and, when a process group is active, the
flat gradient bucket is all-reduced over RCCL before the step.
"""
from typing import Callable

import numpy as np
import torch

from ..optim import FusedAdam
from .trainer import _EarlyScalar
from ..utils import get_array_memsize, set_train_rng


class viBaseTrainer:
    def __init__(self):
        set_train_rng(1)
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.in_dim = None
        self.out_dim = None
        self.z_dim = 1
        self.encoder_net = None
        self.decoder_net = None
        self.train_iterator = None
        self.test_iterator = None
        self.aux_model_params = []
        self.optim = None
        self.current_epoch = 0
        self.metadict = {}
        self.loss_history = {"train_loss": [], "test_loss": []}
        self.filename = "model"
        self.training_cycles = 1
        self.batch_size = 1
        self.dp = None

    def set_model(self, encoder_net, decoder_net) -> None:
        self.encoder_net = encoder_net.to(self.device)
        self.decoder_net = decoder_net.to(self.device)

    def set_encoder(self, encoder_net) -> None:
        self.encoder_net = encoder_net.to(self.device)

    def set_decoder(self, decoder_net) -> None:
        self.decoder_net = decoder_net.to(self.device)

    def set_data(self, X_train, y_train=None, X_test=None, y_test=None, memory_alloc: float = 4) -> None:
        size = sum(get_array_memsize(x) for x in (X_train, y_train, X_test, y_test))
        on_cpu = (size / 1e9) > memory_alloc
        self.train_iterator = self._set_data(X_train, y_train, on_cpu)
        if X_test is not None:
            self.test_iterator = self._set_data(X_test, y_test, on_cpu)

    def _2torch(self, X, y=None):
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(X).float()
        if isinstance(y, np.ndarray):
            y = torch.from_numpy(y).long()                   # class labels (vae.py:580-592)
        return X, y

    def _set_data(self, X, y=None, store_on_cpu: bool = False):
        if X is None:
            raise AssertionError("You must provide input train/test data")
        dev = 'cpu' if store_on_cpu else self.device
        X, y = self._2torch(X, y)
        tensors = (X.to(dev),) if y is None else (X.to(dev), y.to(dev))
        return torch.utils.data.DataLoader(torch.utils.data.TensorDataset(*tensors),
                                           batch_size=self.batch_size, shuffle=True, drop_last=True)

    def elbo_fn(self):
        raise NotImplementedError

    def forward_compute_elbo(self):
        raise NotImplementedError

    def _reset_rng(self, seed: int) -> None:
        set_train_rng(seed)

    def _reset_training_history(self) -> None:
        self.loss_history = {"train_loss": [], "test_loss": []}

    def _delete_optimizer(self) -> None:
        self.optim = None

    def compile_trainer(self, train_data, test_data=None, optimizer=None, elbo_fn: Callable = None,
                        training_cycles: int = 100, batch_size: int = 32, **kwargs) -> None:
        """Same arguments as the reference (vitrainer.py:173-221); default optimizer Adam(lr=1e-4)."""
        self.training_cycles = training_cycles
        self.batch_size = batch_size
        if elbo_fn is not None:
            self.elbo_fn = elbo_fn
        alloc = kwargs.get("memory_alloc", 4)
        distributed = bool(kwargs.get("distributed", False))
        rank, world = 0, 1
        if distributed:          # rank r trains on its shard, eps drawn per rank
            from .. import parallel
            rank, world, _ = parallel.init_distributed(force=True)
            if world > 1:
                train_data = parallel.shard_train_data(train_data[0], train_data[1], rank, world)
        if test_data is not None:
            self.set_data(*train_data, *test_data, memory_alloc=alloc)
        else:
            self.set_data(*train_data, memory_alloc=alloc)
        params = list(self.decoder_net.parameters()) + list(self.encoder_net.parameters())
        for aux in self.aux_model_params:
            params.extend(list(aux))
        if self.optim is None:
            self.optim = FusedAdam(params, lr=1e-4) if optimizer is None else optimizer(params)
        if isinstance(self.optim, FusedAdam):
            self.optim.prepare()
        if distributed and self.dp is None:
            if not isinstance(self.optim, FusedAdam):
                raise TypeError("distributed=True needs the flat-bucket optimizer (FusedAdam)")
            from ..parallel import DataParallelGrads, offset_rng_by_rank
            self.dp = DataParallelGrads(self.optim)
            offset_rng_by_rank(rank)
        self.filename = kwargs.get("filename", "./model")

    @classmethod
    def reparameterize(cls, z_mean: torch.Tensor, z_sd: torch.Tensor) -> torch.Tensor:
        """z = mean + sd * eps with eps drawn on the compute device (vitrainer.py:223-234)."""
        eps = z_mean.new(z_mean.size(0), z_mean.size(1)).normal_()
        return z_mean + z_sd * eps

    def _unpack(self, batch):
        if len(batch) == 1:
            return batch[0].to(self.device), None
        return batch[0].to(self.device), batch[1].to(self.device)

    def train_epoch(self):
        self.decoder_net.train()
        self.encoder_net.train()
        c, elbo_epoch = 0, 0
        for batch in self.train_iterator:
            x, y = self._unpack(batch)
            b = x.size(0)
            elbo = self.forward_compute_elbo(x) if y is None else self.forward_compute_elbo(x, y)
            early = _EarlyScalar(elbo)             # the value travels to the host while backward + Adam still run
            (-elbo).backward()
            if self.dp is not None:
                self.dp.allreduce_grads()
            self.optim.step()
            self.optim.zero_grad()
            elbo = early.item()
            c += b
            elbo_epoch += b * (elbo - elbo_epoch) / c
        return elbo_epoch

    def evaluate_model(self):
        self.decoder_net.eval()
        self.encoder_net.eval()
        c, elbo_epoch = 0, 0
        for batch in self.test_iterator:
            x, y = self._unpack(batch)
            b = x.size(0)
            elbo = (self.forward_compute_elbo(x, mode="eval") if y is None
                    else self.forward_compute_elbo(x, y, mode="eval")).item()
            c += b
            elbo_epoch += b * (elbo - elbo_epoch) / c
        return elbo_epoch

    def print_statistics(self, e):
        if self.dp is not None and self.dp.rank != 0:
            return
        msg = 'Epoch: {}/{}, Training loss: {:.4f}'.format(e + 1, self.training_cycles,
                                                           -self.loss_history["train_loss"][-1])
        if self.test_iterator is not None:
            msg += ', Test loss: {:.4f}'.format(-self.loss_history["test_loss"][-1])
        print(msg)

    def save_model(self, *args: str) -> None:
        savepath = args[0] if args else self.filename
        self.metadict["encoder"] = self.encoder_net.state_dict()
        self.metadict["decoder"] = self.decoder_net.state_dict()
        self.metadict["optimizer"] = self.optim.as_torch_adam() if isinstance(self.optim, FusedAdam) else self.optim
        if self.dp is None or self.dp.rank == 0:
            torch.save(self.metadict, savepath + ".tar")

    def save_weights(self, *args: str) -> None:
        savepath = args[0] if args else self.filename + "weights"
        torch.save({"encoder": self.encoder_net.state_dict(), "decoder": self.decoder_net.state_dict()},
                   savepath + ".tar")

    def load_weights(self, filepath: str) -> None:
        weights = torch.load(filepath, map_location=self.device)
        self.encoder_net.load_state_dict(weights["encoder"])
        self.encoder_net.eval()
        self.decoder_net.load_state_dict(weights["decoder"])
        self.decoder_net.eval()
