"""Checkpoint loaders (reference: atomai/models/loaders.py:25-195, 236-292).

The ``*_metadict_final.tar`` files written by the trainers (trainers/trainer.py ``save_model`` here,
atomai/trainers/trainer.py:344-358 and atomai/trainers/vitrainer.py:361-377 in the reference) are plain
``torch.save`` pickles of {architecture kwargs, weights, optimizer}.  Both directions interchange: a file
written by the reference loads here onto the HIP modules (same state-dict keys / shapes), and files written
here hold only torch types (the fused optimizer is stored as its ``torch.optim.Adam`` equivalent).

Model families outside SURVEY.md section 8 (imspec / reg / cls / denoising autoencoder / joint VAEs) raise.
"""
import warnings
from copy import deepcopy as dc
from typing import Dict, Tuple, Type, Union

import torch

from ..utils import average_weights
from .dgm import VAE, BaseVAE, rVAE
from .segmentor import Segmentor

_OUT_OF_SCOPE = ("imspec", "reg", "cls", "denoising_autoencoder")


def _read(filepath: str) -> Dict:
    device = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.load(filepath, map_location=device, weights_only=False)


def load_model(filepath: str) -> Union[Segmentor, BaseVAE, Dict[str, torch.Tensor]]:
    """Rebuilds a trained model (evaluation state) from a meta-state dictionary file."""
    loaded = _read(filepath)
    if "model_type" not in loaded:
        warnings.warn("Returning model's state dictionary. You will need to load it into your model's "
                      "skeleton by yourself", UserWarning)
        return loaded["weights"]
    model_type = loaded.pop("model_type")
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", category=UserWarning)
        if model_type == "seg":
            return load_seg_model(loaded)
        if model_type == "vae":
            return load_vae_model(loaded)
    if model_type in _OUT_OF_SCOPE:
        raise NotImplementedError(f"model type '{model_type}' is outside the MI355X hot path of this build")
    raise ValueError("The model type {} cannot be loaded".format(model_type))


def load_seg_model(meta_dict: Dict) -> Type[Segmentor]:
    """Segmentor from {model, nb_classes, weights, [optimizer], **architecture kwargs}."""
    name = meta_dict.pop("model")
    nb_classes = meta_dict.pop("nb_classes")
    weights = meta_dict.pop("weights")
    model = Segmentor(name, nb_classes, **meta_dict)
    model.net.load_state_dict(weights)
    if "optimizer" in meta_dict:
        model.optimizer = meta_dict.pop("optimizer")
    model.net.eval()
    return model


def load_vae_model(meta_dict: Dict) -> Type[BaseVAE]:
    """VAE / rVAE from {in_dim, latent_dim, coord, encoder, decoder, optimizer, **kwargs}."""
    in_dim = meta_dict.pop("in_dim")
    latent_dim = meta_dict.pop("latent_dim")
    enc_w, dec_w = meta_dict.pop("encoder"), meta_dict.pop("decoder")
    coord = meta_dict.pop("coord")
    optimizer = meta_dict.pop("optimizer")
    if meta_dict.get("discrete_dim"):
        raise NotImplementedError("joint (discrete) VAEs are outside the MI355X hot path of this build")
    if coord:
        m = rVAE(in_dim, latent_dim, translation=(coord == 3), **meta_dict)
    else:
        m = VAE(in_dim, latent_dim, **meta_dict)
    m.encoder_net.load_state_dict(enc_w)
    m.encoder_net.eval()
    m.decoder_net.load_state_dict(dec_w)
    m.decoder_net.eval()
    m.optim = optimizer
    return m


def load_ensemble(filepath: str) -> Tuple[Type[torch.nn.Module], Dict[int, Dict[str, torch.Tensor]]]:
    """Single model with the ensemble-averaged weights + the dictionary of all members' weights."""
    loaded = _read(filepath)
    if "model_type" not in loaded:
        warnings.warn("Returning dictionary with ensemble weights. You will need to load them into your "
                      "model's skeleton by yourself")
        return None, dc(loaded["weights"])
    model_type = loaded.pop("model_type")
    ensemble_weights = dc(loaded["weights"])
    loaded["weights"] = average_weights(loaded["weights"])
    if model_type == "seg":
        smodel = load_seg_model(loaded)
    elif model_type in _OUT_OF_SCOPE:
        raise NotImplementedError(f"model type '{model_type}' is outside the MI355X hot path of this build")
    else:
        raise ValueError("The model type {} cannot be loaded".format(model_type))
    return smodel.net, ensemble_weights


def load_pretrained_model(model_name: str):
    """'G_MD' / 'BFO' pretrained Segmentors.  The reference downloads them (loaders.py:274-292); this build
    runs without network access, so the ``.tar`` must already be in the working directory."""
    import os
    fname = {"BFO": "./bfo.tar", "G_MD": "./G_MD.tar"}.get(model_name)
    if fname is None:
        raise ValueError("Available pretrained models are 'G_MD' and 'BFO'")
    if not os.path.exists(fname):
        raise FileNotFoundError(f"{fname} not found: download it from the reference's 'pretrained' folder")
    return load_model(fname)
