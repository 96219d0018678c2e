"""numpy -> tensor plumbing for Segmentor training / prediction
(reference: atomai/utils/preproc.py:18-74, 138-278, 365-421, 798-825)."""
import warnings

import numpy as np
import torch


def num_classes_from_labels(labels: np.ndarray) -> int:
    """Number of classes from the label values: must be 0..K-1; two values mean ONE class (binary)."""
    uval = np.unique(labels)
    if min(uval) != 0:
        raise AssertionError("Labels should start from 0")
    if np.any(np.diff(uval) != 1):
        raise AssertionError("Mask values should be in range between 0 and total number of classes "
                             "with an increment of 1")
    k = len(uval)
    return k - 1 if k == 2 else k


def check_image_dims(X_train, y_train, X_test, y_test, num_classes: int):
    """Adds the channel axis to 3-D image stacks, and to 3-D masks in the single-class case."""
    def chan(a, what):
        if a.ndim == 3:
            warnings.warn(f'Adding a channel dimension of 1 to {what}', UserWarning)
            return a[:, None]
        return a
    X_train, X_test = chan(X_train, "training images"), chan(X_test, "test images")
    if num_classes == 1:
        y_train, y_test = chan(y_train, "training labels"), chan(y_test, "test labels")
    return X_train, y_train, X_test, y_test


def get_array_memsize(X_arr, precision: str = "single") -> float:
    """Bytes the array takes once cast to single (or double) precision."""
    if X_arr is None:
        return 0
    if precision not in ("single", "double"):
        raise NotImplementedError("Specify 'single' or 'double' precision type")
    n = X_arr.numel() if isinstance(X_arr, torch.Tensor) else X_arr.size
    return n * (4 if precision == "single" else 8)


def _data_device(store_on_cpu: bool) -> str:
    return 'cuda' if torch.cuda.is_available() and not store_on_cpu else 'cpu'


def array2list_(x, batch_size: int, store_on_cpu: bool = False):
    if not isinstance(x, (np.ndarray, torch.Tensor)):
        raise TypeError("Provide data as numpy array or torch tensor")
    if isinstance(x, torch.Tensor):
        x = x.to(_data_device(store_on_cpu))
    n_batches = x.shape[0] // batch_size
    split = np.split if isinstance(x, np.ndarray) else torch.chunk
    return split(x[:n_batches * batch_size], n_batches)


def array2list(X_train, y_train, X_test, y_test, batch_size: int, memory_alloc: float = 4):
    """Lists of whole mini-batches (remainder dropped); device-resident if the set is < memory_alloc GB."""
    data = [X_train, y_train, X_test, y_test]
    on_cpu = sum(get_array_memsize(x) for x in data) / 1e9 > memory_alloc
    return tuple(array2list_(x, batch_size, on_cpu) for x in data)


def preprocess_training_image_data_(images_all, labels_all, images_test_all, labels_test_all):
    data = (images_all, labels_all, images_test_all, labels_test_all)
    all_np = all(isinstance(i, np.ndarray) for i in data)
    all_t = all(isinstance(i, torch.Tensor) for i in data)
    if not all_np and not all_t:
        raise TypeError("Provide training and test data in the form of numpy arrays or torch tensors")
    num_classes = num_classes_from_labels(labels_all if all_np else labels_all.cpu().numpy())
    X, y, Xt, yt = check_image_dims(*data, num_classes)
    if all_np:
        X, y, Xt, yt = (torch.from_numpy(np.ascontiguousarray(a)) for a in (X, y, Xt, yt))
    X, Xt = X.float(), Xt.float()
    y, yt = (y.long(), yt.long()) if num_classes > 1 else (y.float(), yt.float())
    return X, y, Xt, yt, num_classes


def preprocess_training_image_data(images_all, labels_all, images_test_all, labels_test_all,
                                   batch_size: int, memory_alloc: float = 4):
    *tensors, num_classes = preprocess_training_image_data_(
        images_all, labels_all, images_test_all, labels_test_all)
    return (*array2list(*tensors, batch_size, memory_alloc), num_classes)


def init_dataloaders(X_train, y_train, X_test, y_test, batch_size: int, memory_alloc: float = 4):
    on_cpu = sum(get_array_memsize(x) for x in (X_train, y_train, X_test, y_test)) / 1e9 > memory_alloc
    dev = _data_device(on_cpu)
    tr = torch.utils.data.TensorDataset(X_train.to(dev), y_train.to(dev))
    te = torch.utils.data.TensorDataset(X_test.to(dev), y_test.to(dev))
    return (torch.utils.data.DataLoader(tr, batch_size=batch_size, shuffle=True, drop_last=True),
            torch.utils.data.DataLoader(te, batch_size=batch_size, drop_last=True))


def init_fcnn_dataloaders(X_train, y_train, X_test, y_test, batch_size: int, num_classes=None,
                          memory_alloc: float = 4):
    *tensors, num_classes = preprocess_training_image_data_(X_train, y_train, X_test, y_test)
    return (*init_dataloaders(*tensors, batch_size, memory_alloc), num_classes)


def torch_format_image(image_data: np.ndarray, norm: bool = True) -> torch.Tensor:
    """(n,h,w) -> float32 (n,1,h,w); global min-max normalisation over the WHOLE stack."""
    if image_data.ndim not in (3, 4):
        raise AssertionError("Provide image(s) as 3D (n, h, w) or 4D (n, 1, h, w) tensor")
    if image_data.ndim == 3:
        image_data = image_data[:, None]
    if norm:
        image_data = (image_data - image_data.min()) / np.ptp(image_data)
    return torch.from_numpy(np.ascontiguousarray(image_data)).float()


def to_onehot(idx: torch.Tensor, n: int) -> torch.Tensor:
    """One-hot encoding of integer labels (reference: atomai/utils/preproc.py:915-928; the reference allocates on
    'cuda' whenever CUDA exists, here the result lives on the labels' own device)."""
    if torch.max(idx).item() >= n:
        raise AssertionError("Labelling must start from 0 and "
                             "maximum label value must be less than total number of classes")
    if idx.dim() == 1:
        idx = idx.unsqueeze(1)
    onehot = torch.zeros(idx.size(0), int(n), device=idx.device)
    onehot.scatter_(1, idx.long(), 1)
    return onehot
