"""Coordinate grids for the rVAE spatial decoder (reference: atomai/utils/coords.py:37-83)."""
from typing import Tuple, Union

import numpy as np
import torch


def grid2xy(X1: torch.Tensor, X2: torch.Tensor) -> torch.Tensor:
    """(M, N) grids -> (M*N, 2) xy pairs."""
    return torch.stack((X1.reshape(-1), X2.reshape(-1)), 1)


def imcoordgrid(im_dim: Tuple) -> torch.Tensor:
    """x in linspace(-1, 1, H) along rows, y in linspace(1, -1, W) along columns, 'ij' meshgrid."""
    xx = torch.linspace(-1, 1, im_dim[0])
    yy = torch.linspace(1, -1, im_dim[1])
    x0, x1 = torch.meshgrid(xx, yy, indexing="ij")
    return grid2xy(x0, x1)


def transform_coordinates(coord: Union[np.ndarray, torch.Tensor], phi: torch.Tensor,
                          coord_dx: Union[np.ndarray, torch.Tensor, int] = 0) -> torch.Tensor:
    """Batched rotation by phi (coord @ [[cos, sin], [-sin, cos]]) followed by translation.  B x n x 2
    elementwise math (17 MB at bs 512, 64x64): torch ops, differentiable."""
    if isinstance(coord, np.ndarray):
        coord = torch.from_numpy(coord).float()
    if isinstance(coord_dx, np.ndarray):
        coord_dx = torch.from_numpy(coord_dx).float()
    c, s = torch.cos(phi)[:, None], torch.sin(phi)[:, None]
    x, y = coord[..., 0], coord[..., 1]
    return torch.stack((x * c - y * s, x * s + y * c), -1) + coord_dx
