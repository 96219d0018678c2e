from .imaug import datatransform, seg_augmentor  # noqa: F401

__all__ = ["datatransform", "seg_augmentor"]
