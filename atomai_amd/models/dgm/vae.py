"""BaseVAE / VAE (reference: atomai/models/dgm/vae.py:28-221, 594-747)."""
from copy import deepcopy as dc
from typing import List, Optional, Tuple

import numpy as np
import torch

from ...losses_metrics import vae_loss
from ...nets import init_VAE_nets
from ...trainers import viBaseTrainer
from ...utils import set_train_rng, to_onehot
from ...utils.coords import imcoordgrid


class BaseVAE(viBaseTrainer):
    """Encoder/decoder object shared by the VAE family (vae.py:28-221)."""

    def __init__(self, in_dim: Tuple[int], latent_dim: int, nb_classes: int = 0, coord: int = 0,
                 discrete_dim: Optional[List] = None, seed: int = 0, **kwargs) -> None:
        super().__init__()
        msg = ("You must specify the input dimensions and pass them as a tuple. For images, specify "
               "(height, width) or (height, width, channels) if multiple channels. For spectra, specify (length,)")
        if in_dim is None or not isinstance(in_dim, (tuple, list)):
            raise AssertionError(msg)
        if isinstance(in_dim, tuple) and not isinstance(in_dim[0], int):
            raise AssertionError(msg)
        set_train_rng(seed)              # NB: the nets are always drawn under BaseVAE's own seed (default 0)
        self.in_dim = in_dim
        self.z_dim = latent_dim
        self.discrete_dim = discrete_dim
        if coord:
            if len(in_dim) not in (2, 3):
                raise NotImplementedError("VAE with rotation and translational invariance are available "
                                          "only for 2D image data")
            self.z_dim = self.z_dim + coord
            self.x_coord = imcoordgrid(in_dim).to(self.device)
        self.nb_classes = nb_classes
        encoder_net, decoder_net, self.metadict = init_VAE_nets(in_dim, latent_dim, coord, discrete_dim,
                                                                nb_classes, **kwargs)
        self.set_model(encoder_net, decoder_net)
        self.sigmoid_out = self.metadict["sigmoid_out"]
        self.coord = coord

    def encode_(self, x_new, **kwargs) -> np.ndarray:
        """Encoder forward in ``num_batches`` chunks -> concatenated (z_mean | z_logsd) array."""
        if isinstance(x_new, np.ndarray):
            x_new = torch.from_numpy(x_new).float()
        if x_new.ndim == len(self.in_dim):
            x_new = x_new.unsqueeze(0)
        x_new = x_new.to(self.device)
        num_batches = kwargs.get("num_batches", 10)
        bs = max(1, len(x_new) // num_batches)
        self.encoder_net.eval()
        out = []
        with torch.no_grad():
            for i in range(0, len(x_new), bs):
                out.append(torch.cat(self.encoder_net(x_new[i:i + bs]), -1).cpu().numpy())
        return np.concatenate(out)

    def encode(self, x_new, **kwargs) -> Tuple[np.ndarray]:
        z = self.encode_(x_new, **kwargs)
        return z[:, :self.z_dim], z[:, self.z_dim:]

    def decode(self, z_sample, y=None) -> np.ndarray:
        """Maps latent point(s) to data space with the trained generative model (vae.py:178-221)."""
        if isinstance(z_sample, np.ndarray):
            z_sample = torch.from_numpy(z_sample).float()
        if z_sample.dim() == 1:
            z_sample = z_sample[None, ...]
        z_sample = z_sample.to(self.device)
        if y is not None:                                    # class-conditioned decoding (vae.py:200-209)
            if isinstance(y, int):
                y = torch.tensor(y)
            elif isinstance(y, np.ndarray):
                y = torch.from_numpy(y)
            if y.dim() == 0:
                y = y.unsqueeze(0)
            z_sample = torch.cat((z_sample, to_onehot(y.to(self.device), self.nb_classes)), dim=-1)
        self.decoder_net.to(self.device).eval()
        with torch.no_grad():
            if self.coord:
                x_coord = self.x_coord.expand(z_sample.size(0), *self.x_coord.size()).contiguous()
                x_decoded = self.decoder_net(x_coord, z_sample)
            else:
                x_decoded = self.decoder_net(z_sample)
        if self.sigmoid_out:
            x_decoded = torch.sigmoid(x_decoded)
        return x_decoded.cpu().numpy()

    def reconstruct(self, x_new, **kwargs) -> np.ndarray:
        """Decodes ``num_samples`` draws from the encoded distribution of ONE input (vae.py:223-271; regular VAE:
        the coordinate latents are dropped)."""
        num_samples = kwargs.get("num_samples", 32)
        label = kwargs.get("label")
        z_mean, z_sd = self.encode(x_new)
        z_mean = torch.from_numpy(z_mean[:, self.coord:])
        z_sd = torch.from_numpy(z_sd[:, self.coord:])
        ndist = torch.distributions.Normal(z_mean, torch.exp(z_sd))
        alphas = None if label is None else to_onehot(torch.tensor(label).unsqueeze(0), self.nb_classes)
        out = []
        for _ in range(num_samples):
            z_sample = ndist.rsample().view(1, -1)
            if alphas is not None:                           # class to be reconstructed (vae.py:255-267)
                z_sample = torch.cat([z_sample, alphas], dim=1)
            out.append(self.decode(z_sample))
        return np.concatenate(out, axis=0)

    def encode_image_(self, img: np.ndarray, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """Encodes the training-window-sized sub-image around EVERY pixel of a 2-D image (vae.py:300-344); returns the
        image and its latent map, both cropped to the pixels whose window fits."""
        from ...utils import crop_borders, extract_subimages, get_coord_grid
        num_batches = kwargs.get("num_batches", 10)
        inf = int(1e5)
        img_to_encode = img.copy()
        coordinates = get_coord_grid(img_to_encode, 1, return_dict=False)
        batch_size = coordinates.shape[0] // num_batches
        encoded_img = -inf * np.ones((*img_to_encode.shape, self.z_dim))
        bounds = [(i * batch_size, (i + 1) * batch_size) for i in range(num_batches)]
        bounds.append((num_batches * batch_size, coordinates.shape[0]))
        for lo, hi in bounds:
            coord_i = coordinates[lo:hi]
            if len(coord_i) == 0:
                continue
            subimgs_i, com_i, _ = extract_subimages(img_to_encode, coord_i, self.in_dim[0])
            if len(subimgs_i) > 0:
                z_mean, _ = self.encode(subimgs_i, num_batches=10)
                encoded_img[com_i[:, 0].astype(int), com_i[:, 1].astype(int)] = z_mean
        img_to_encode[encoded_img[..., 0] == -inf] = 0
        img_to_encode = crop_borders(img_to_encode[..., None], 0)
        encoded_img = crop_borders(encoded_img, -inf)
        return img_to_encode[..., 0], encoded_img

    def encode_images(self, imgdata: np.ndarray, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """``encode_image_`` for every image of a stack (vae.py:273-298)."""
        if (imgdata.ndim == len(self.in_dim) == 2 or imgdata.ndim == len(self.in_dim) == 3):
            imgdata = np.expand_dims(imgdata, axis=0)
        imgs, encoded = [], []
        for i, img in enumerate(imgdata):
            print("\rImage {}/{}".format(i + 1, imgdata.shape[0]), end="")
            img_, enc_ = self.encode_image_(img, **kwargs)
            imgs.append(img_)
            encoded.append(enc_)
        return np.array(imgs), np.array(encoded)

    def _check_inputs(self, X_train, y_train=None, X_test=None, y_test=None) -> None:
        if self.in_dim != X_train.shape[1:]:
            raise RuntimeError("The values of input dimensions you specified do not match "
                               "the training data dimensions")
        if X_test is not None and self.in_dim != X_test.shape[1:]:
            raise RuntimeError("The values of input dimensions you specified do not match "
                               "the test data dimensions")
        if y_train is not None and self.nb_classes == 0:     # vae.py:563-578
            raise RuntimeError("You must have forgotten to specify number of classes during the initialization. "
                               "Example of correct usage: vae = VAE(in_dim=(28, 28), nb_classes=10)); "
                               "vae.fit(train_data, train_labels).")
        lbl_match = True
        if y_train is not None and y_test is None:
            lbl_match = self.nb_classes == len(np.unique(y_train))
        elif y_train is not None and y_test is not None:
            lbl_match = self.nb_classes == len(np.unique(y_train)) == len(np.unique(y_test))
        if not lbl_match:
            raise RuntimeError("The number of classes specified at initialization must be "
                               "equal the the number of classes in train and test labels")

    def update_metadict(self):
        self.metadict["num_epochs"] = self.current_epoch
        self.metadict["num_iter"] = self.kdict_["num_iter"]

    def _fit_loop(self):
        for e in range(self.training_cycles):
            self.current_epoch = e
            self.loss_history["train_loss"].append(self.train_epoch())
            if self.test_iterator is not None:
                self.loss_history["test_loss"].append(self.evaluate_model())
            self.print_statistics(e)
            self.update_metadict()
            self.save_model(self.filename)


class VAE(BaseVAE):
    """Plain variational autoencoder (vae.py:594-747)."""

    def __init__(self, in_dim: int = None, latent_dim: int = 2, nb_classes: int = 0, seed: int = 0,
                 **kwargs) -> None:
        super().__init__(in_dim, latent_dim, nb_classes, 0, **kwargs)
        set_train_rng(seed)
        self.kdict_ = dc(kwargs)
        self.kdict_["num_iter"] = 0
        self.loss = "mse"

    def elbo_fn(self, x, x_reconstr, *args, **kwargs) -> torch.Tensor:
        return vae_loss(self.loss, self.in_dim, x, x_reconstr, *args, **kwargs)

    def forward_compute_elbo(self, x: torch.Tensor, y=None, mode: str = "train") -> torch.Tensor:
        x = x.to(self.device)
        with torch.set_grad_enabled(mode != "eval"):
            z_mean, z_logsd = self.encoder_net(x)
            if mode != "eval":
                self.kdict_["num_iter"] += 1
            z = self.reparameterize(z_mean, torch.exp(z_logsd))
            if y is not None:                                # vae.py:677-680
                z = torch.cat((z, to_onehot(y, self.nb_classes)), -1)
            x_reconstr = self.decoder_net(z)
            return self.elbo_fn(x, x_reconstr, z_mean, z_logsd, **self.kdict_)

    def fit(self, X_train, y_train=None, X_test=None, y_test=None, loss: str = "mse", **kwargs) -> None:
        self._check_inputs(X_train, y_train, X_test, y_test)
        for k, v in kwargs.items():
            if k in ["capacity"]:
                self.kdict_[k] = v
        self.compile_trainer((X_train, y_train), (X_test, y_test), **kwargs)
        self.loss = loss
        if self.loss == "ce":                                 # decode() then applies a sigmoid ("prediction" stage)
            self.sigmoid_out = True
            self.metadict["sigmoid_out"] = True
        self._fit_loop()
