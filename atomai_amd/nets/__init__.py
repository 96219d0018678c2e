from .blocks import ConvBlock, DilatedBlock, UpsampleBlock
from .fcnn import Unet, dilnet, init_fcnn_model

__all__ = ["ConvBlock", "UpsampleBlock", "DilatedBlock", "Unet", "dilnet", "init_fcnn_model"]
