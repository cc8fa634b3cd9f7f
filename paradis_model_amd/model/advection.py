"""Neural semi-Lagrangian advection (drop-in for reference ``model/advection.py:7-175``).

Constructor signature, child names (``down_projection`` / ``up_projection``) and the
non-persistent geometry buffers match the reference; the forward runs

    down-projection (stencil + MFMA GEMM)  ->  ONE fused HIP kernel for
    [pole mean, departure point, virtual geocyclic halo, bilinear/bicubic gather, pole mean]
    ->  up-projection (MFMA GEMM)

instead of ~30 ATen passes; backward recomputes the departure point from (field, u, v).
"""
import torch

from .. import ops
from .blocks import GMBlock
from .padding import GeoCyclicPadding


class NeuralSemiLagrangian(torch.nn.Module):
    """Neural semi-Lagrangian advection operator."""

    def __init__(self, cfg, hidden_dim: int, mesh_size: tuple, num_vels: int,
                 lat_grid: torch.Tensor, lon_grid: torch.Tensor, interpolation: str = "bicubic"):
        super().__init__()
        self.padding = 2 if interpolation == "bicubic" else 1
        self.padding_interp = GeoCyclicPadding(self.padding)
        self.hidden_dim = hidden_dim
        self.num_vels = num_vels
        self.mesh_size = mesh_size
        self.interpolation = interpolation

        adv_cfg = cfg.model.physblock.advection
        self.down_projection = GMBlock(layers=adv_cfg.down_projection.layers, input_dim=hidden_dim,
                                       output_dim=num_vels, mesh_size=mesh_size,
                                       hidden_dim=adv_cfg.down_projection.hidden_dim)
        self.up_projection = GMBlock(layers=adv_cfg.up_projection.layers, input_dim=num_vels,
                                     output_dim=hidden_dim, mesh_size=mesh_size,
                                     hidden_dim=adv_cfg.up_projection.hidden_dim)

        H, W = mesh_size
        buf = lambda name, t: self.register_buffer(name, t, persistent=False)  # noqa: E731
        buf("lat_grid", lat_grid.unsqueeze(0).unsqueeze(0).contiguous().clone())
        buf("lon_grid", lon_grid.unsqueeze(0).unsqueeze(0).contiguous().clone())
        buf("Hf", torch.tensor(float(H)))
        buf("Wf", torch.tensor(float(W)))
        buf("min_lat", torch.min(lat_grid))
        buf("max_lat", torch.max(lat_grid))
        buf("min_lon", torch.min(lon_grid))
        buf("max_lon", torch.max(lon_grid))
        buf("d_lon", self.max_lon - self.min_lon)
        buf("d_lat", self.max_lat - self.min_lat)
        # host-side tables (sin/cos of the arrival latitude, longitudes) for the fused kernel
        self._geom = ops.AdvectGeometry(lat_grid, lon_grid)

    def advect(self, projected: torch.Tensor, u: torch.Tensor, v: torch.Tensor, dt: float):
        """The fused core on already projected planes [B, num_vels, H, W]."""
        return ops.sl_advect(projected, u, v, self._geom, dt, self.interpolation)

    def forward(self, hidden_features: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                dt: float) -> torch.Tensor:
        projected = self.down_projection(hidden_features)
        interpolated = self.advect(projected, u, v, dt)
        return self.up_projection(interpolated)

    def forward_velocities(self, hidden_features: torch.Tensor, velocities: torch.Tensor,
                           dt: float, return_skip: bool = False):
        """Same as ``forward`` with u = velocities[:, :K], v = velocities[:, K:] passed as one tensor
        (what ``Paradis._layer_step`` has at hand): saves the slice/zero-fill/copy passes of autograd.
        ``return_skip``: also hand back ``hidden_features`` for the caller's gated blend (``(advected, hidden)``), so
        the blend's gradient enters the down-projection's first backward kernel (``GMBlock.forward``)."""
        if return_skip:
            projected, skip = self.down_projection(hidden_features, return_skip=True)
        else:
            projected = self.down_projection(hidden_features)
        interpolated = ops.sl_advect_vel(projected, velocities, self._geom, dt, self.interpolation)
        out = self.up_projection(interpolated)
        return (out, skip) if return_skip else out

    def transport(self, hidden_features: torch.Tensor, velocities: torch.Tensor, dt: float,
                  alpha: torch.Tensor) -> torch.Tensor:
        """``h + sigmoid(alpha) * (A(h) - h)`` (reference model/paradis.py:236-243: the advection followed by the
        caller's gated blend) in one chain: the blend runs in the epilogue of the up-projection's last GEMM, so the
        advected tensor is never written, and the blend's gradient of ``h`` enters the down-projection's first
        backward kernel.  Bit-identical to ``ops.gated_blend(h, self.forward_velocities(h, ...), alpha)``."""
        projected, skip = self.down_projection(hidden_features, return_skip=True)
        interpolated = ops.sl_advect_vel(projected, velocities, self._geom, dt, self.interpolation)
        return self.up_projection(interpolated, residual=skip, gate=alpha)


# north_star spelling
SemiLagrangianAdvection = NeuralSemiLagrangian
