"""Drop-in ``model`` package: same module/class names as the reference's ``model/`` package
(``model.paradis.Paradis``, ``model.advection.NeuralSemiLagrangian``, ``model.padding.GeoCyclicPadding``,
``model.blocks.*``), with the compute in hand-written gfx950 HIP kernels."""
from .padding import GeoCyclicPadding, GeocyclicPadding
from .blocks import (BLOCK_REGISTRY, ChannelNorm, CLinear, GlobalBias, GMBlock, PhysicalDownsample,
                     SepConv, init_conv2d_default, init_module_convs)
from .advection import NeuralSemiLagrangian, SemiLagrangianAdvection
from .paradis import Paradis, get_scaled_timestep

__all__ = ["GeoCyclicPadding", "GeocyclicPadding", "BLOCK_REGISTRY", "ChannelNorm", "CLinear",
           "GlobalBias", "GMBlock", "PhysicalDownsample", "SepConv", "init_conv2d_default",
           "init_module_convs", "NeuralSemiLagrangian", "SemiLagrangianAdvection", "Paradis",
           "get_scaled_timestep"]
