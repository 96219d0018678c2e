"""dklGPTrainer: deep-kernel-learning GP training loop (reference: atomai/trainers/gptrainer.py:144-349): the
shared-embedding path (``compile_trainer``) and the independent per-output networks / ensembles
(``compile_multi_model_trainer``)."""
from typing import Tuple

import numpy as np
import torch

import copy

from ..nets.gp import GPModelList, GPRegressionModel, fcFeatureExtractor


class dklGPTrainer:
    def __init__(self, indim: int, embedim: int = 2, shared_embedding_space: bool = True, **kwargs) -> None:
        seed = kwargs.get("seed", 42)
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        self.dimdict = {"input_dim": indim, "embedim": embedim}
        # the CURRENT device (not a hard-coded cuda:0): DKL runs as independent replicas, one per rank / GPU
        self.device = kwargs.get("device", f'cuda:{torch.cuda.current_device()}' if torch.cuda.is_available() else 'cpu')
        precision = kwargs.get("precision", "double")
        # explicit dtype/device instead of the reference's global torch.set_default_tensor_type
        self.dtype = torch.float32 if precision == "single" else torch.float64
        self.correlated_output = shared_embedding_space
        self.ensemble = False
        # GP layer: "kissgp" = the reference's GridInterpolationKernel model (gp.py:41-46), "exact" = dense exact GP
        self.gp_kind = kwargs.get("gp", "kissgp")
        self.gp_model = None
        self.likelihood = None
        self.compiled = False
        self.train_loss = []

    def _set_data(self, x, device: str = None) -> torch.Tensor:
        dev = device if device else self.device
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        elif not isinstance(x, torch.Tensor):
            raise TypeError("Pass data as ndarray or torch tensor object")
        return x.to(self.dtype).to(dev)

    def set_data(self, x, y=None, device: str = None) -> Tuple[torch.Tensor]:
        x = self._set_data(x, device)
        if y is not None:
            y = y[None] if y.ndim == 1 else y
            y = self._set_data(y, device)
        return x, y

    def _build_extractor(self, feature_net, input_dim: int, embedim: int):
        """The reference makes the chosen precision the process-wide default tensor type (utils/nn.py:149-167), so
        under precision='double' — the trainer's default — the Linear layers are DRAWN in float64 (a different
        consumption of the generator than float32 draws cast up).  Same here, scoped to the construction and on the
        CPU generator for any device (pinned by tests/golden/gp_extractor.npz)."""
        prev = torch.get_default_dtype()
        torch.set_default_dtype(self.dtype)
        try:
            net = feature_net(input_dim, embedim)
        finally:
            torch.set_default_dtype(prev)
        return net.to(self.dtype).to(self.device)

    def compile_multi_model_trainer(self, X, y, training_cycles: int = 1, **kwargs) -> None:
        """One feature extractor + one GP PER OUTPUT (gptrainer.py:181-243): for vector-valued targets with independent
        latent spaces every model starts from a copy of the same initial network; in ensemble mode (``self.ensemble``,
        set by ``dklGPR.fit_ensemble``) every member draws its own initialisation.  Adam(lr=0.01) over all parameters,
        loss = - sum of the members' marginal log likelihoods."""
        if self.correlated_output:
            raise NotImplementedError("To compile a DKL-GP trainer for correlated outputs "
                                      "use compile_trainer(*args, **kwargs)")
        X, y = self.set_data(X, y)
        if y.shape[0] < 2:
            raise ValueError("The training targets must be vector-valued (d >1)")
        input_dim, embedim = self.dimdict["input_dim"], self.dimdict["embedim"]
        feature_net = kwargs.get("feature_extractor", fcFeatureExtractor)
        freeze = kwargs.get("freeze_weights", False)

        def new_extractor():
            fx = self._build_extractor(feature_net, input_dim, embedim)
            if freeze:
                for p in fx.parameters():
                    p.requires_grad = False
            return fx
        shared_init = None if self.ensemble else new_extractor()
        models = []
        for i in range(y.shape[0]):
            fx = new_extractor() if self.ensemble else copy.deepcopy(shared_init)
            models.append(GPRegressionModel(X, y[i:i + 1], fx, embedim, kwargs.get("base_kernel", "rbf"),
                                            kwargs.get("grid_size", 50), self.gp_kind))
        self.gp_model = GPModelList(models).to(self.device)
        self.likelihood = self.gp_model
        self.gp_model.train()
        params = []
        for m in self.gp_model.models:
            params += [m.raw_lengthscale, m.raw_outputscale, m.mean_constant, m.raw_noise]
            if not freeze:
                params += list(m.feature_extractor.parameters())
        self.optimizer = self._make_optimizer([{'params': params}], 0.01)
        self.training_cycles = training_cycles
        self.compiled = True

    def _make_optimizer(self, groups, lr: float):
        """Adam over the reference's parameter groups (gptrainer.py:289-296: one learning rate for all of them).  In
        single precision on the device the groups are ONE flat bucket of the fused optimizer (optim.FusedAdam: one HIP
        launch per step, the convolutional extractor's weight gradients written straight into the bucket by the tape);
        double precision — the reference's default — keeps torch.optim.Adam: adam.hip is an fp32 kernel."""
        params = [p for g in groups for p in g['params']]
        if self.dtype == torch.float32 and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            from ..optim import FusedAdam
            opt = FusedAdam(params, lr=lr)
            opt.prepare()
            return opt
        return torch.optim.Adam(groups, lr=lr)

    def compile_trainer(self, X, y, training_cycles: int = 1, **kwargs) -> None:
        """feature extractor NN + base kernel + Adam(lr=0.01) over kernel, mean, noise (+ NN) parameters."""
        if not self.correlated_output:
            raise NotImplementedError("To compile a DKL-GP trainer for independent outputs use "
                                      "compile_multi_model_trainer(*args, **kwargs)")
        X, y = self.set_data(X, y)
        input_dim, embedim = self.dimdict["input_dim"], self.dimdict["embedim"]
        feature_net = kwargs.get("feature_extractor", fcFeatureExtractor)
        feature_extractor = self._build_extractor(feature_net, input_dim, embedim)
        freeze = kwargs.get("freeze_weights", False)
        if freeze:
            for p in feature_extractor.parameters():
                p.requires_grad = False
        self.gp_model = GPRegressionModel(X, y, feature_extractor, embedim, kwargs.get("base_kernel", "rbf"),
                                          kwargs.get("grid_size", 50), self.gp_kind)
        self.gp_model.to(self.device)
        self.likelihood = self.gp_model          # the Gaussian noise lives in the same module (raw_noise)
        self.gp_model.train()
        m = self.gp_model
        groups = [{'params': [m.raw_lengthscale, m.raw_outputscale]}, {'params': [m.mean_constant]},
                  {'params': [m.raw_noise]}]
        if not freeze:
            groups.append({'params': list(m.feature_extractor.parameters())})
        self.optimizer = self._make_optimizer(groups, kwargs.get("lr", 0.01))
        self.training_cycles = training_cycles
        self.compiled = True

    def train_step(self) -> None:
        self.optimizer.zero_grad()
        loss = -self.gp_model.mll()
        loss.backward()
        self.optimizer.step()
        self.train_loss.append(loss.item())

    def run(self, X=None, y=None, training_cycles: int = 1, **kwargs):
        if not self.compiled:
            if self.correlated_output:
                self.compile_trainer(X, y, training_cycles, **kwargs)
            else:
                self.compile_multi_model_trainer(X, y, training_cycles, **kwargs)
        for e in range(self.training_cycles):
            self.train_step()
            if e == 0 or (e + 1) % kwargs.get("print_loss", 10) == 0 or e == self.training_cycles - 1:
                self.print_statistics(e)
        return self.gp_model

    def print_statistics(self, e):
        print('Epoch {}/{} ...'.format(e + 1, self.training_cycles),
              'Training loss: {}'.format(np.around(self.train_loss[-1], 4)))

    def save_weights(self, filename: str) -> None:
        """Saves the feature extractor weights only, as the reference does (gptrainer.py:347-349)."""
        torch.save(self.gp_model.feature_extractor.state_dict(), filename)
