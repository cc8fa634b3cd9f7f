"""ParadisLoss mirror (reference ``utils/loss.py:7-282``; SURVEY.md row f1, "next").

Same constructor arguments and weighting rules as the reference, including its
blocks-of-``num_levels`` walk over the first ``num_features-num_surface_vars`` channels.
On the HIP device ``forward`` is ONE fused kernel (``paradis_loss_fwd_bwd``: weighted loss value and
d loss / d pred in a single pass over pred/target); on CPU tensors (weight construction tests, the
gloo harness test) it is the plain elementwise formula.
"""
import re

import torch


class ParadisLoss(torch.nn.Module):
    def __init__(self, loss_function: str, lat_grid: torch.Tensor, pressure_levels: torch.Tensor,
                 num_features: int, num_surface_vars: int, var_loss_weights: torch.Tensor,
                 output_name_order: list, delta_loss: float = 1.0,
                 apply_latitude_weights: bool = False) -> None:
        super().__init__()
        if loss_function not in ("mse", "reversed_huber"):
            raise Exception(f"{loss_function} not supported, choose between [reversed_huber, mse]")
        self.kind = loss_function
        self.pressure_levels = pressure_levels.to(torch.float32)
        self.delta = delta_loss
        self.num_levels = len(pressure_levels)
        self.num_features = num_features
        self.num_surface_vars = num_surface_vars
        self.num_atmospheric_vars = num_features - num_surface_vars
        self.var_loss_weights = var_loss_weights
        self.output_name_order = output_name_order
        self.apply_latitude_weights = apply_latitude_weights
        self.lat_weights = self._latitude_weights(lat_grid)
        self.register_buffer("lat_weights_buf", self.lat_weights.view(1, 1, -1, 1), persistent=False)
        self.feature_weights = self._feature_weights()
        self.register_buffer("feature_weights_buf", self.feature_weights.view(1, -1, 1, 1),
                             persistent=False)

    @staticmethod
    def _latitude_weights(lat_deg: torch.Tensor) -> torch.Tensor:
        lat = lat_deg.to(torch.float64)
        if lat.ndim != 1:
            raise ValueError(f"grid_lat_deg must be 1D [H], got {lat.shape}")
        steps = lat[1:] - lat[:-1]
        if not torch.allclose(steps, steps[0].expand_as(steps), rtol=0.0, atol=1e-6):
            raise ValueError("Latitude grid is not uniformly spaced.")
        delta = steps[0].abs()
        lo, hi = float(lat.min()), float(lat.max())
        if abs(lo + 90.0) <= 1e-6 and abs(hi - 90.0) <= 1e-6:      # grid holds both poles
            w = torch.cos(torch.deg2rad(lat)) * torch.sin(torch.deg2rad(delta) / 2.0)
            w[torch.argmin(lat)] = w[torch.argmax(lat)] = torch.sin(torch.deg2rad(delta) / 4.0) ** 2
        else:
            half = float(delta) / 2.0
            if abs(hi - (90.0 - half)) > 1e-6 or abs(lo - (-90.0 + half)) > 1e-6:
                raise ValueError(f"Latitude vector must end at ±(90 - Δ/2). Got min={lo}, max={hi}.")
            w = torch.cos(torch.deg2rad(lat))
        return (w / w.mean()).to(dtype=lat_deg.dtype)

    def _feature_weights(self) -> torch.Tensor:
        plev = self.pressure_levels / 1000
        pw = torch.where(plev > 0.2, plev, torch.full_like(plev, 0.2))
        fw = torch.zeros(self.num_features, dtype=torch.float32)
        for i in range(0, self.num_atmospheric_vars, self.num_levels):
            _ = re.sub(r"_h\d+$", "", self.output_name_order[i])
            fw[i:i + self.num_levels] = self.var_loss_weights[i:i + self.num_levels] * pw
        fw[self.num_atmospheric_vars:] = self.var_loss_weights[self.num_atmospheric_vars:]
        return fw

    # The two methods below are the reference's per-variable validation metric (utils/loss.py:105-127): a [C] vector for
    # logging, off the training hot path, a handful of ATen elementwise ops on whatever device the tensors live on.
    # The training loss itself (forward) has one implementation only, the fused HIP kernel.
    def _pointwise_loss(self, pred, target):
        if self.kind == "mse":
            return (pred - target) ** 2
        d = self.delta
        err = pred - target
        mag = err.abs()
        blend = 1 / (1 + torch.exp(-2 * (mag - d)))
        return (1 - blend) * (d * mag) + blend * ((err ** 2 + d ** 2) / (2 * d))

    def per_channel_loss(self, pred, target, weighted: bool = True):
        loss = self._pointwise_loss(pred, target)
        if weighted:
            loss = loss * self.feature_weights_buf
            if self.apply_latitude_weights:
                loss = loss * self.lat_weights_buf
        return loss.mean(dim=(0, 2, 3))

    def forward(self, pred, target):
        """Fused HIP kernel (value and d/dpred in one pass); like every module of this package it refuses CPU tensors -
        the CPU evaluation of the loss is ``oracle.paradis_oracle.paradis_loss`` (tests only)."""
        from . import ops
        lw = self.lat_weights_buf.reshape(-1) if self.apply_latitude_weights else None
        return ops.paradis_loss(pred, target, self.feature_weights_buf.reshape(-1), lw, self.kind, self.delta)


def build_loss(cfg, lat_deg: torch.Tensor) -> ParadisLoss:
    """Loss-weight assembly of reference ``trainer.py:112-187`` from the config."""
    from .config import feature_layout
    lay = feature_layout(cfg)
    vw = cfg.training.variable_loss_weights
    weights = torch.zeros(lay.num_out_features, dtype=torch.float32)
    for i, feat in enumerate(lay.output_name_order):
        base = re.sub(r"_h\d+$", "", feat)
        if base in vw.atmospheric:
            weights[i] = vw.atmospheric[base]
        elif base in vw.surface:
            weights[i] = vw.surface[base]
        else:
            raise ValueError(f"No loss weight configured for output feature '{feat}' "
                             f"(base variable '{base}').")
    return ParadisLoss(loss_function=cfg.training.loss_function.type, lat_grid=lat_deg,
                       pressure_levels=torch.tensor(cfg.features.pressure_levels, dtype=torch.float32),
                       num_features=lay.num_out_features,
                       num_surface_vars=len(cfg.features.output.surface), var_loss_weights=weights,
                       output_name_order=lay.output_name_order,
                       delta_loss=cfg.training.loss_function.delta_loss,
                       apply_latitude_weights=cfg.training.loss_function.lat_weights)
