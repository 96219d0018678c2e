"""Dev-container-only harness that imports the *real* reference (pycroscopy/atomai,
mounted read-only at /root/reference) so golden vectors can be generated from it.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file, and it
cannot work on the GPU box (/root/reference does not exist there).  It is used by
``oracle/make_golden.py`` (fixture generator) and by the ``not gpu`` tests that pin the
oracle against the live reference when it is present.

The reference does not import on this image as-is:
  * cv2 / skimage / torchvision / mendeleev / gpytorch / progressbar / ase are absent
    -> served as auto-stub modules by a meta-path finder;
  * ``torch.utils.data.TensorDataset()`` with zero tensors raises on torch 2.10
    (atomai/trainers/trainer.py:90-91) -> __init__ shim;
  * ``np.product`` (atomai/losses_metrics/vi_losses.py:28, the 'ce' reconstruction loss) was removed in
    numpy 2 -> the alias ``np.product = np.prod`` is restored for the reference's benefit.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_STUB_ROOTS = {"cv2", "skimage", "torchvision", "mendeleev", "gpytorch",
               "progressbar", "ase", "linear_operator"}


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if name[:1].isupper():
            obj = type(name, (_Dummy,), {"__module__": self.__name__})
        else:
            obj = sys.modules.get(full)
            if obj is None:
                obj = _StubModule(full)
                sys.modules[full] = obj
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "atomai"))


_installed = False


def import_reference():
    """Returns the imported reference package ``atomai`` (CPU only)."""
    global _installed
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    if not _installed:
        import matplotlib
        matplotlib.use("Agg")
        import torch
        sys.meta_path.insert(0, _Finder())
        _orig = torch.utils.data.TensorDataset.__init__

        def _init(self, *tensors):
            if len(tensors) == 0:
                self.tensors = ()
                return
            _orig(self, *tensors)
        torch.utils.data.TensorDataset.__init__ = _init
        import numpy
        if not hasattr(numpy, "product"):
            numpy.product = numpy.prod
        sys.path.insert(0, REFERENCE_ROOT)
        _installed = True
    import atomai
    return atomai
