from .blocks import ConvBlock, DilatedBlock, ResBlock, ResModule, UpsampleBlock
from .ed import convDecoderNet, convEncoderNet, coord_latent, fcDecoderNet, fcEncoderNet, init_VAE_nets, rDecoderNet
from .fcnn import ResHedNet, SegResNet, Unet, dilnet, init_fcnn_model
from .gp import GPRegressionModel, convFeatureExtractor, fcFeatureExtractor

__all__ = ["ConvBlock", "UpsampleBlock", "DilatedBlock", "ResBlock", "ResModule", "Unet", "dilnet", "SegResNet", "ResHedNet",
           "init_fcnn_model",
           "fcEncoderNet", "convEncoderNet", "convDecoderNet", "fcDecoderNet", "rDecoderNet", "coord_latent", "init_VAE_nets",
           "fcFeatureExtractor", "convFeatureExtractor", "GPRegressionModel"]
