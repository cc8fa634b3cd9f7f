"""Simple ADR building blocks (drop-in for reference ``model/blocks.py``).

Same class names, constructor keywords, child-module names and parameter shapes as the
reference, so state dicts load unchanged and optimiser param-grouping by module type
(``nn.Conv2d`` / ``nn.Linear`` holders, reference ``trainer.py:24-64``) behaves the same.
The modules only *own* parameters; all arithmetic runs in the gfx950 HIP kernels of
``paradis_model_amd.ops``:

* ``CLinear`` / pointwise half of ``SepConv`` -> FP32-MFMA GEMM with fused bias, GlobalBias map,
  activation and residual epilogue;
* depthwise half of ``SepConv`` -> LDS-tiled stencil on the *virtual* geocyclic halo;
* ``ChannelNorm`` -> per-pixel channel reduction kernel (unbiased variance);
* ``GlobalBias`` -> rank-128 separable map kernel; inside a ``GMBlock`` the map is added in the
  GEMM epilogue instead of a separate pass.
"""
from collections import OrderedDict
from collections.abc import Sequence
from typing import Optional, Tuple, Type, Union

import torch
from torch import nn

from .. import ops
from .padding import GeoCyclicPadding

_ACT_TYPES = (nn.SiLU, nn.GELU)


def init_conv2d_default(conv: nn.Conv2d, *, scale: float = 1.0) -> None:
    """He-normal (fan-in, relu gain) weights, optional down-scaling, zero bias
    (reference model/blocks.py:33-39)."""
    nn.init.kaiming_normal_(conv.weight, mode="fan_in", nonlinearity="relu")
    if scale != 1.0:
        with torch.no_grad():
            conv.weight.mul_(scale)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0.0)


def init_module_convs(m: nn.Module, *, last_conv_scale: float = 1.0) -> None:
    """Re-initialise every ``nn.Conv2d`` under ``m`` in traversal order; the last one is scaled
    (reference model/blocks.py:42-54)."""
    convs = [mod for mod in m.modules()
             if not isinstance(mod, GlobalBias) and isinstance(mod, nn.Conv2d)]
    for i, conv in enumerate(convs):
        init_conv2d_default(conv, scale=last_conv_scale if i == len(convs) - 1 else 1.0)


class PhysicalDownsample(nn.Module):
    """Geocyclic 5x5 box filter + decimation by ``stride`` (reference model/blocks.py:57-71).
    ``stride=1`` is a blur, not the identity."""

    def __init__(self, stride=4):
        super().__init__()
        self.stride = stride
        # attribute names of the reference, kept for introspection; the HIP kernel fuses both
        self.pool = nn.AvgPool2d(kernel_size=5, stride=stride, count_include_pad=False)
        self.padding = GeoCyclicPadding(2)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.avgpool_geo(x, self.stride)


# diagnostic switch: False materialises the projected GlobalBias map (the pre-fusion path)
FUSE_BIAS_PROJECTION = True
# below this plane size the 8 MB map stays in L2 and the two paths tie (measured 128.9 vs 128.7
# samples/s at 32x64); at 721x1440 the map is 4.25 GB per GlobalBias and fusing is +10.6 %
FUSE_BIAS_PROJECTION_MIN_POINTS = 8192


class CLinear(nn.Module):
    """Channel-wise linear map = per-sample GEMM (reference model/blocks.py:74-89)."""

    def __init__(self, input_dim: int, output_dim: int, mesh_size: tuple, kernel_size: int = 1,
                 bias: bool = True):
        super().__init__()
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size=1, bias=bias)

    def forward(self, x, bias_map=None, act: Optional[str] = None, residual=None, x_pre=None,
                x_act=None, defer_act_grad=False, bias_proj=None, gate=None, out_bf16: bool = False):
        return ops.pointwise(x, self.conv.weight, self.conv.bias, bias_map, residual, act, x_pre, x_act,
                             defer_act_grad, bias_proj, gate=gate, out_bf16=out_bf16)


class SepConv(nn.Module):
    """Depthwise k x k on the geocyclic halo, then pointwise (reference model/blocks.py:92-116)."""

    def __init__(self, input_dim: int, output_dim: int, mesh_size: tuple, kernel_size: int = 3,
                 bias: bool = True):
        super().__init__()
        if kernel_size % 2 == 0 or not 1 <= kernel_size <= 11:
            # (the reference's own padding arithmetic, (k-1)//2 per side, only preserves the grid for odd k)
            raise NotImplementedError("SepConv: the HIP depthwise stencil supports odd kernel_size 1..11; "
                                      f"got {kernel_size}")
        self.padding = (kernel_size - 1) // 2
        self.geo_padding = GeoCyclicPadding(self.padding)
        self.depthwise = nn.Conv2d(input_dim, input_dim, kernel_size, groups=input_dim, bias=False)
        self.pointwise = nn.Conv2d(input_dim, output_dim, kernel_size=1, bias=bias)

    def forward(self, x, bias_map=None, act: Optional[str] = None, residual=None, bias_proj=None,
                with_skip: bool = False, gate=None):
        """``with_skip``: also hand back the input for a consumer around the block (``(out, x)``): its gradient is
        then added inside the stencil's data-gradient kernel (``ops.dwconv_geo_skip``)."""
        skip = None
        # (the stencil's output has one consumer, the pointwise GEMM below: a bf16 tensor inside autocast(bfloat16))
        if with_skip:
            x, skip = ops.dwconv_geo_skip(x, self.depthwise.weight, self.depthwise.bias, out_bf16=True)
        else:
            x = ops.dwconv_geo(x, self.depthwise.weight, self.depthwise.bias, out_bf16=True)
        out = ops.pointwise(x, self.pointwise.weight, self.pointwise.bias, bias_map, residual, act,
                            bias_proj=bias_proj, gate=gate)
        return (out, skip) if with_skip else out


class ChannelNorm(nn.Module):
    """Per-pixel normalisation over channels, unbiased variance, eps 1e-5
    (reference model/blocks.py:118-134)."""

    def __init__(self, input_dim: int, output_dim: int):
        super().__init__()
        assert input_dim == output_dim
        self.eps = 1e-5
        self.weight = nn.Parameter(torch.ones(input_dim), requires_grad=True)
        self.bias = nn.Parameter(torch.zeros(input_dim), requires_grad=True)

    def forward(self, x, x_extra=None, with_skip: bool = False, out_bf16: bool = False):
        """``x_extra``: optional second tensor treated as concatenated after ``x`` along channels
        (the reaction block's cat([hidden, hidden_static]) without materialising it).
        ``with_skip``: also return ``x`` for a residual branch; its gradient is then added inside
        the backward kernel (no separate accumulation pass).
        ``out_bf16``: the only consumer of the output is a pointwise layer - a request honoured inside
        ``torch.autocast(bfloat16)`` only (``ops._want_bf16_out``): the output comes back as a bf16 tensor."""
        if with_skip:
            return ops.channel_norm_skip(x, self.weight, self.bias, self.eps, x_extra, out_bf16=out_bf16)
        return ops.channel_norm(x, self.weight, self.bias, self.eps, x_extra, out_bf16=out_bf16)


class GlobalBias(nn.Module):
    """Low-rank separable bias map  y_c = sum_k A[c,k] U[k,:] V[k,:]^T, optionally projected to
    ``output_dim`` channels (reference model/blocks.py:138-197)."""

    def __init__(self, input_dim: int, output_dim: int, *, bias: bool = True, kernel_size: int = 0,
                 mesh_size: Tuple[int, int], rank: int = 128):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.rank = rank
        self.height, self.width = mesh_size
        self.A = nn.Parameter(torch.zeros(input_dim, rank), requires_grad=True)
        self.U = nn.Parameter(torch.zeros(rank, self.height), requires_grad=True)
        self.V = nn.Parameter(torch.zeros(rank, self.width), requires_grad=True)
        with torch.no_grad():
            nn.init.normal_(self.A, mean=0.0, std=1e-3)
            nn.init.normal_(self.U, mean=0.0, std=1e-3)
            nn.init.normal_(self.V, mean=0.0, std=1e-3)
        self.projection = (nn.Linear(input_dim, output_dim, bias=False)
                           if input_dim != output_dim else None)

    def bias_terms(self):
        """(bias_map, bias_proj) for the fused GEMM epilogue: with a projection (<= 16 bias channels,
        output_dim a multiple of 4) the [output_dim,H,W] map is never materialised - the GEMM adds
        sum_c P[o,c] * m8[c,h,w] on the fly."""
        big = self.U.shape[1] * self.V.shape[1] >= FUSE_BIAS_PROJECTION_MIN_POINTS
        if (FUSE_BIAS_PROJECTION and big and self.projection is not None and self.input_dim <= 16
                and self.output_dim % 4 == 0):
            return None, (ops.global_bias_m8(self.A, self.U, self.V), self.projection.weight)
        return self.bias_map(), None

    def bias_map(self) -> torch.Tensor:
        """[output_dim, H, W]; batch independent, recomputed every forward like the reference."""
        pw = self.projection.weight if self.projection is not None else None
        return ops.global_bias_map(self.A, self.U, self.V, pw)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.add_bias_map(x, self.bias_map())


BLOCK_REGISTRY = {
    "SepConv": SepConv,
    "CLinear": CLinear,
    "ChannelNorm": ChannelNorm,
    "GlobalBias": GlobalBias,
}


class GMBlock(nn.Sequential):
    """Generic multilayer block (reference model/blocks.py:210-304): optional ChannelNorm, then
    ``layers`` with a GlobalBias after the first layer and an activation after every layer but the
    last (unless ``activation=True``).  Child names follow the reference (``"{idx}-{Type}"``).

    ``forward`` walks the children and fuses  [layer, GlobalBias?, activation?]  into one GEMM with
    epilogue; ``residual`` (added after the last layer) and ``x_extra`` (virtual channel concat in
    front of the leading ChannelNorm) are extensions used by ``Paradis._layer_step``.
    """

    def __init__(self, layers: Sequence[Union[str, Type[nn.Module]]], input_dim: int, output_dim: int,
                 mesh_size: Tuple[int, int], kernel_size: Union[Sequence[int], int] = 5,
                 hidden_dim: Union[Sequence, int] = 0, activation_fn: Type[nn.Module] = nn.SiLU,
                 bias_channels: int = 0, activation: Union[Sequence, bool] = False,
                 pre_normalize: bool = False):
        depth = len(layers)
        if depth == 0:
            raise ValueError("GMBlock: must specify at least one layer")

        def per_layer(value, count, what):
            """scalar -> one entry per slot; a sequence must already have one entry per slot"""
            if isinstance(value, Sequence) and not isinstance(value, str):
                if len(value) != count:
                    raise AssertionError(f"GMBlock: {what} needs {count} entries, got {len(value)}")
                return tuple(value)
            return (value,) * count

        # an activation follows every layer but the last; `activation` (bool) decides the last one
        activation = (per_layer(activation, depth, "activation") if isinstance(activation, Sequence)
                      else (True,) * (depth - 1) + (bool(activation),))
        # widths between layers: `hidden_dim` <= 0 means "as wide as the wider end of the block"
        if not isinstance(hidden_dim, Sequence) and hidden_dim <= 0:
            hidden_dim = max(input_dim, output_dim)
        hidden_dim = per_layer(hidden_dim, depth - 1, "hidden_dim")
        kernel_size = per_layer(kernel_size, depth, "kernel_size")
        num_layers = depth

        children = []
        if pre_normalize:
            children.append(("0-ChannelNorm", ChannelNorm(input_dim=input_dim, output_dim=input_dim)))
        width_in = input_dim
        for idx, spec in enumerate(layers):
            if isinstance(spec, str):
                if spec not in BLOCK_REGISTRY:
                    raise ValueError(
                        f"Unknown layer type: {spec}. Available: {list(BLOCK_REGISTRY.keys())}")
                cls = BLOCK_REGISTRY[spec]
            else:
                cls = spec
            width_out = output_dim if idx == num_layers - 1 else hidden_dim[idx]
            children.append((f"{idx}-{cls.__name__}",
                             cls(input_dim=width_in, output_dim=width_out, mesh_size=mesh_size,
                                 kernel_size=kernel_size[idx])))
            if idx == 0 and bias_channels > 0:
                children.append(("0-GlobalBias", GlobalBias(input_dim=bias_channels,
                                                            output_dim=width_out,
                                                            mesh_size=mesh_size)))
            if activation[idx]:
                children.append((f"{idx}-{activation_fn.__name__}", activation_fn()))
            width_in = width_out

        super().__init__(OrderedDict(children))
        init_module_convs(self, last_conv_scale=0.1)

    def forward(self, x, residual=None, x_extra=None, return_skip: bool = False, gate=None):
        """``return_skip``: also return the block input for other consumers (``(out, x)``); with a
        leading ChannelNorm the gradients of all those consumers are summed inside its backward
        kernel.  The same happens automatically when ``residual`` is the block input itself.
        ``gate`` (with ``residual``): the block output is BLENDED with the residual per channel,
        ``residual + sigmoid(gate) * (block(x) - residual)`` (reference model/paradis.py:239-243), inside the last
        layer's GEMM epilogue."""
        if gate is not None and residual is None:
            raise ValueError("GMBlock: a gate needs the residual it blends with")
        mods = list(self.children())
        n = len(mods)
        i = 0
        skip, x_in = None, x
        fuse_skip = (n > 0 and isinstance(mods[0], ChannelNorm) and torch.is_grad_enabled()
                     and x.requires_grad and (return_skip or residual is x))
        # a block that starts with a depthwise stencil: the other consumers' gradients enter its data-gradient kernel
        stencil_skip = (n > 0 and isinstance(mods[0], SepConv) and torch.is_grad_enabled() and x.requires_grad
                        and return_skip and residual is None and x_extra is None)
        if x_extra is not None and not (n and isinstance(mods[0], ChannelNorm)):
            x = torch.cat([x, x_extra], dim=1)
            x_extra = None
        pre = pre_act = None        # pre-activation hand-off between two chained CLinear layers
        while i < n:
            m = mods[i]
            if isinstance(m, ChannelNorm):
                to_gemm = i + 1 < n and isinstance(mods[i + 1], CLinear)     # sole consumer: a pointwise GEMM
                if i == 0 and fuse_skip:
                    res_is_x = residual is x
                    x, skip = m(x, x_extra, with_skip=True, out_bf16=to_gemm)
                    if res_is_x:
                        residual = skip
                else:
                    x = m(x, x_extra, out_bf16=to_gemm)
                x_extra = None
                i += 1
            elif isinstance(m, (CLinear, SepConv)):
                j = i + 1
                bias_map = bias_proj = act = None
                if j < n and isinstance(mods[j], GlobalBias):
                    bias_map, bias_proj = mods[j].bias_terms()
                    j += 1
                if j < n and isinstance(mods[j], _ACT_TYPES):
                    act = type(mods[j]).__name__
                    j += 1
                res = residual if j == n else None
                # the activation gradient moves into the next layer's dgrad epilogue when that
                # layer is a CLinear (only consumer of this output) and autograd is recording
                hand_off = (act is not None and res is None and j < n and isinstance(mods[j], CLinear)
                            and isinstance(m, CLinear) and torch.is_grad_enabled())
                g = gate if res is not None else None
                # the sole consumer of this activated output is the next CLinear: under autocast(bfloat16) it is
                # handed over as a bf16 TENSOR, as the reference's conv2d returns it there (a request; ops.pointwise
                # honours it in the bf16-mixed scheme only)
                chain16 = (act is not None and res is None and j < n and isinstance(mods[j], CLinear)
                           and isinstance(m, CLinear))
                if isinstance(m, CLinear):
                    out = m(x, bias_map=bias_map, act=act, residual=res, x_pre=pre, x_act=pre_act,
                            defer_act_grad=hand_off, bias_proj=bias_proj, gate=g, out_bf16=chain16)
                elif i == 0 and stencil_skip:
                    out, skip = m(x, bias_map=bias_map, act=act, residual=res, bias_proj=bias_proj, with_skip=True)
                else:
                    out = m(x, bias_map=bias_map, act=act, residual=res, bias_proj=bias_proj, gate=g)
                if hand_off:
                    x, pre = out
                    pre_act = act
                else:
                    x, pre, pre_act = out, None, None
                if res is not None:
                    residual = None
                i = j
            elif isinstance(m, _ACT_TYPES):
                x = ops.activation(x, type(m).__name__)
                i += 1
            else:
                x = m(x)
                i += 1
        if residual is not None:
            x = ops.add(x, residual) if gate is None else ops.gated_blend(residual, x, gate)
        if return_skip:
            return x, (skip if skip is not None else x_in)
        return x
