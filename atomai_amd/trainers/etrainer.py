"""Deep-ensemble trainers (reference: atomai/trainers/etrainer.py:29-512).

Same orchestration (train_baseline / train_ensemble_from_scratch / train_ensemble_from_baseline / train_swag),
attribute names and ``*_ensemble_metadict.tar`` format; every member trains on the HIP engine through
``BaseTrainer``.  Only the segmentation families are in scope ('imspec' raises).

Multi-GPU (no counterpart in the reference): members are independent training runs, so under an
initialised process group ``distributed=True`` gives rank r the members r, r + world, ... on its own GPU — no
collective on the training path (and no gradient all-reduce: every member sees the full batch) — and ONE
``all_gather_object`` of the finished members' CPU state dicts at the end, so that every rank returns the complete
ensemble; rank 0 writes the metadict.  ``member_range`` selects an explicit slice instead.
"""
import warnings
from copy import deepcopy as dc
from typing import Callable, Dict, Tuple, Type, Union

import numpy as np
import torch
import torch.distributed as dist

from ..nets import init_fcnn_model
from ..utils import (average_weights, check_image_dims, num_classes_from_labels, sample_weights)
from .trainer import BaseTrainer

ensemble_type = Dict[int, Dict[str, torch.Tensor]]


class BaseEnsembleTrainer(BaseTrainer):
    def __init__(self, model: Type[torch.nn.Module] = None, nb_classes=None) -> None:
        super().__init__()
        if model is not None:
            self.set_model(model, nb_classes)
        self.ensemble_state_dict = {}
        self.kdict = {}
        self._ens_rank = 0                                   # rank inside a sharded ensemble run (0 writes the file)

    # ------------------------------------------------------------------ sharding over ranks
    def _members(self, n_models: int, kwargs: dict):
        """Member indices this process trains, and whether they were sharded over the process group."""
        distributed = kwargs.pop("distributed", False)
        member_range = kwargs.pop("member_range", None)
        members = list(range(n_models) if member_range is None else member_range)
        self._ens_rank = 0
        if distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.dp is not None:
                raise AssertionError("a sharded ensemble trains independent members: do not combine it with "
                                     "data-parallel gradient averaging (trainer.dp)")
            self._ens_rank = dist.get_rank()
            return members[self._ens_rank::dist.get_world_size()], True
        return members, False

    def _gather_members(self) -> None:
        """Every rank receives every member (CPU tensors); the only collective of a sharded ensemble run."""
        local = {i: {k: v.detach().cpu() for k, v in sd.items()} for i, sd in self.ensemble_state_dict.items()}
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
        merged = {}
        for part in parts:
            merged.update(part)
        dev = self.device
        self.ensemble_state_dict = {i: {k: v.to(dev) for k, v in merged[i].items()} for i in sorted(merged)}

    def compile_ensemble_trainer(self, **kwargs) -> None:
        """kwargs are forwarded to BaseTrainer.compile_trainer for every member."""
        self.kdict = kwargs

    def train_baseline(self, X_train, y_train, X_test=None, y_test=None, seed: int = 1, augment_fn=None):
        if self.net is None:
            raise AssertionError("You need to set a model first")
        self._reset_rng(seed)
        self._reset_weights()
        self._reset_training_history()
        self._delete_optimizer()
        X_train, y_train, X_test, y_test = self.preprocess_train_data(X_train, y_train, X_test, y_test)
        self.compile_trainer((X_train, y_train, X_test, y_test), **self.kdict)
        self.data_augmentation(augment_fn)
        self.fit()
        return self.net

    def train_ensemble_from_scratch(self, X_train, y_train, X_test=None, y_test=None, n_models: int = 10,
                                    augment_fn=None, **kwargs) -> Tuple[Type[torch.nn.Module], ensemble_type]:
        """Every member starts from a different initialisation (seed = batch_seed = member index)."""
        members, sharded = self._members(n_models, kwargs)
        self.update_training_parameters(kwargs)
        print("Training ensemble models (strategy = 'from_scratch')")
        for i in members:
            print("\nEnsemble model {}".format(i + 1))
            self.kdict["batch_seed"] = i
            model_i = self.train_baseline(X_train, y_train, X_test, y_test, i, augment_fn)
            self.ensemble_state_dict[i] = dc(model_i.state_dict())
            self.save_ensemble_metadict()
        if sharded:
            self._gather_members()
            self.save_ensemble_metadict()
        return self.net, self.ensemble_state_dict

    def train_ensemble_from_baseline(self, X_train, y_train, X_test=None, y_test=None, basemodel=None,
                                     n_models: int = 10, training_cycles_base: int = 1000,
                                     training_cycles_ensemble: int = 100, augment_fn=None, **kwargs):
        """Members continue from a common baseline with different batch shuffling (seed i + 2).  Sharded runs train
        the (deterministic) baseline redundantly on every rank instead of broadcasting it."""
        members, sharded = self._members(n_models, kwargs)
        self.update_training_parameters(kwargs)
        if basemodel is None:
            self.kdict["training_cycles"] = training_cycles_base
            print("Training baseline model...")
            basemodel = self.train_baseline(X_train, y_train, X_test, y_test, 1, augment_fn)
        else:
            X_train, y_train, X_test, y_test = self.preprocess_train_data(X_train, y_train, X_test, y_test)
        self.set_model(basemodel)
        basemodel_state_dict = dc(self.net.state_dict())
        self.kdict["training_cycles"] = training_cycles_ensemble
        if not self.full_epoch and "print_loss" not in self.kdict:
            self.kdict["print_loss"] = 10
        print("\nTraining ensemble models (strategy = 'from_baseline')")
        model_i = self.net
        for j, i in enumerate(members):
            print("\nEnsemble model {}".format(i + 1))
            if j > 0:
                self.net.load_state_dict(basemodel_state_dict)
            self._reset_rng(i + 2)
            self._reset_training_history()
            self._delete_optimizer()
            self.compile_trainer((X_train, y_train, X_test, y_test), batch_seed=i + 2, **self.kdict)
            model_i = self.run()
            self.ensemble_state_dict[i] = dc(model_i.state_dict())
            self.save_ensemble_metadict()
            model_i.load_state_dict(average_weights(self.ensemble_state_dict))
        if sharded:
            self._gather_members()
            self.save_ensemble_metadict()
            model_i.load_state_dict(average_weights(self.ensemble_state_dict))
        return model_i, self.ensemble_state_dict

    def train_swag(self, X_train, y_train, X_test=None, y_test=None, n_models: int = 10, augment_fn=None, **kwargs):
        """SWAG-like sampling of weights from the running weights of a single (SWA) training run."""
        self.update_training_parameters(kwargs)
        self.kdict["swa"] = True
        basemodel = self.train_baseline(X_train, y_train, X_test, y_test, 1, augment_fn)
        self.ensemble_state_dict = sample_weights(self.running_weights, n_models)
        self.save_ensemble_metadict()
        return basemodel, self.ensemble_state_dict

    def update_training_parameters(self, kwargs) -> None:
        msg = "Overwriting the initial value '{}' of parameter '{}' with new value '{}'"
        for k, v in kwargs.items():
            if k in self.kdict:
                warnings.warn(msg.format(self.kdict[k], k, v), UserWarning)
            self.kdict[k] = v

    def preprocess_train_data(self, *train_data):
        return tuple(torch.from_numpy(x) for x in train_data)

    def save_ensemble_metadict(self, filename: str = None) -> None:
        fname = self.filename if filename is None else filename
        meta = dict(self.meta_state_dict)                    # same keys as the reference's file (incl. 'optimizer')
        meta["weights"] = self.ensemble_state_dict
        if (self.dp is None or self.dp.rank == 0) and self._ens_rank == 0:
            torch.save(meta, fname + "_ensemble_metadict.tar")


class EnsembleTrainer(BaseEnsembleTrainer):
    """``EnsembleTrainer('Unet'|'dilnet'|'SegResNet'|'ResHedNet', nb_classes, **net_kwargs)``."""

    def __init__(self, model: Union[str, Type[torch.nn.Module]] = None, nb_classes: int = 1, **kwargs) -> None:
        super().__init__()
        self.nb_classes = nb_classes
        if isinstance(model, str):
            if model in ["Unet", "dilnet", "SegResNet", "ResHedNet"]:
                self.net, self.meta_state_dict = init_fcnn_model(model, self.nb_classes, **kwargs)
                self.accuracy_fn = accuracy_fn_seg(nb_classes)
            elif model == "imspec":
                raise NotImplementedError("the ImSpec family is outside the MI355X hot path of this build")
            else:
                raise NotImplementedError("Currently implemented models are 'Unet', 'dilnet', SegResNet', "
                                          "and 'ResHedNet'")
            self.net.to(self.device)
        else:
            self.set_model(model, nb_classes)
        self.meta_state_dict["weights"] = self.net.state_dict()
        self.meta_state_dict["optimizer"] = self.optimizer

    def compile_ensemble_trainer(self, **kwargs) -> None:
        self.kdict = kwargs
        self.full_epoch = self.kdict.get("full_epoch", False)
        self.batch_size = self.kdict.get("batch_size", 32)
        self.kdict["overwrite_train_data"] = False

    def train_baseline(self, X_train, y_train, X_test=None, y_test=None, seed: int = 1, augment_fn=None):
        if self.net is None:
            raise AssertionError("You need to set a model first")
        train_data = self.preprocess_train_data(X_train, y_train, X_test, y_test)
        self.set_data(*train_data, **self.kdict)
        self._reset_rng(seed)
        self._reset_weights()
        self._reset_training_history()
        self._delete_optimizer()
        self.compile_trainer((X_train, y_train, X_test, y_test), **self.kdict)
        self.data_augmentation(augment_fn)
        self.fit()
        return self.net

    def preprocess_train_data(self, *args):
        if self.meta_state_dict.get("model_type") == "seg":
            return set_data_seg(*args, self.nb_classes)
        raise NotImplementedError("only segmentation ensembles are on the MI355X hot path of this build")


def set_data_seg(X_train, y_train, X_test=None, y_test=None, nb_classes_set: int = 1, **kwargs):
    """Training / test arrays for semantic segmentation (etrainer.py:437-468)."""
    nb_classes = num_classes_from_labels(y_train)
    if nb_classes != nb_classes_set:
        raise AssertionError("Number of specified classes is different from the number of classes "
                             "contained in training data")
    if X_test is None or y_test is None:
        from sklearn.model_selection import train_test_split
        X_train, X_test, y_train, y_test = train_test_split(
            X_train, y_train, test_size=kwargs.get("test_size", .15), shuffle=True,
            random_state=kwargs.get("seed", 1))
    X_train, y_train, X_test, y_test = check_image_dims(X_train, y_train, X_test, y_test, nb_classes)
    X_train, X_test = X_train.astype(np.float32), X_test.astype(np.float32)
    lab = np.int64 if nb_classes > 1 else np.float32
    return X_train, y_train.astype(lab), X_test, y_test.astype(lab)


def accuracy_fn_seg(nb_classes: int) -> Callable:
    def accuracy(y, y_prob, *args):
        raise NotImplementedError("IoU (cv2-based, CPU) is outside the MI355X hot path of this build")
    return accuracy
