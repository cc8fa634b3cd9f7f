"""Geocyclic halo padding (drop-in for reference ``model/padding.py:4-39``)."""
import torch

from .. import ops


class GeoCyclicPadding(torch.nn.Module):
    """Pole-reflecting, longitude-periodic padding of [B,C,H,W] fields on an equiangular grid.

    The halo is produced by one HIP gather kernel through the integer index map
    (bit-exact with the reference's roll/flip/cat construction).  Inside the model the padded
    tensor is never materialised: the stencil and advection kernels index the halo virtually.
    """

    def __init__(self, pad_width):
        super().__init__()
        self.pad_width = pad_width

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.pad_width == 0:
            return x
        return ops.geocyclic_pad(x, self.pad_width)

    def extra_repr(self) -> str:
        return f"pad_width={self.pad_width}"


# north_star spelling
GeocyclicPadding = GeoCyclicPadding
