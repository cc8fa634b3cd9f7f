"""Configuration objects accepted by the drop-in modules.

The reference passes a Hydra/OmegaConf ``DictConfig``; the modules only need
attribute access plus ``.get()`` (reference ``model/paradis.py:34-193`` reads
``cfg.model.*``, ``cfg.compute.get("gradient_checkpointing")``,
``cfg.dataset.n_time_inputs``, ``cfg.features.input.constants``).  ``AttrDict``
provides that without omegaconf; a real ``DictConfig`` works unchanged.

``default_config()`` carries the hot-path keys of the shipped
``config/paradis_settings.yaml`` (values only: latent 1024, 768 velocity planes,
8 layers, bicubic, SiLU, ...), and ``load_yaml`` reads a reference YAML file.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace
from typing import Any


class AttrDict(dict):
    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(obj: Any) -> Any:
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attr(v) for v in obj]
    return obj


PRESSURE_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
_IN_ATM = ["geopotential", "wind_x", "wind_y", "wind_z", "specific_humidity", "temperature"]
_IN_SFC = ["wind_x_10m", "wind_y_10m", "wind_z_10m", "2m_temperature", "mean_sea_level_pressure"]
_FORCINGS = ["toa_incident_solar_radiation", "sin_time_of_day", "cos_time_of_day",
             "sin_year_progress", "cos_year_progress"]
_CONSTANTS = ["geopotential_at_surface", "land_sea_mask", "slope_of_sub_gridscale_orography",
              "standard_deviation_of_orography", "lon_spacing", "cos_latitude", "cos_longitude",
              "sin_longitude", "latitude", "longitude"]


def default_config() -> AttrDict:
    """Hot-path subset of the reference's shipped configuration."""
    pb = {
        "input_proj": {"layers": ["CLinear"], "hidden_dim": 0},
        "velocity_net": {"layers": ["CLinear", "SepConv"], "hidden_dim": 384},
        "diffusion": {"layers": ["SepConv"], "hidden_dim": 0},
        "reaction": {"layers": ["CLinear"] * 4, "hidden_dim": 896},
        "output_proj": {"layers": ["CLinear"] * 3, "hidden_dim": 768},
        "advection": {"down_projection": {"layers": ["SepConv"], "hidden_dim": 0},
                      "up_projection": {"layers": ["CLinear"], "hidden_dim": 0}},
    }
    cfg = {
        "model": {"latent_size": 1024, "forecast_steps": 1, "base_dt": 21600, "num_layers": 8,
                  "bias_channels": 8, "velocity_vectors": 768, "adv_interpolation": "bicubic",
                  "activation": "SiLU", "coarsening_factor": 1, "physblock": pb},
        "init": {"seed": 42},
        "dataset": {"n_time_inputs": 2},
        "compute": {"gradient_checkpointing": False, "compile": False, "use_amp": False},
        "training": {
            "optimizer": {"name": "adamw", "lr": 5e-4, "weight_decay": 1e-2, "beta1": 0.9,
                          "beta2": 0.95, "detach_gradient_every": None},
            "accumulate_grad_batches": 1,
            "loss_function": {"type": "reversed_huber", "delta_loss": 1.0, "lat_weights": True},
            "variable_loss_weights": {
                "atmospheric": {"wind_x": 1.0, "wind_y": 1.0, "wind_z": 1.0, "geopotential": 1.0,
                                "specific_humidity": 1.0, "temperature": 1.0,
                                "vertical_velocity": 0.1},
                "surface": {"wind_x_10m": 1.0, "wind_y_10m": 1.0, "wind_z_10m": 1.0,
                            "2m_temperature": 1.0, "mean_sea_level_pressure": 1.0,
                            "total_precipitation_6hr": 1.0}},
        },
        "features": {
            "pressure_levels": list(PRESSURE_LEVELS),
            "input": {"atmospheric": list(_IN_ATM), "surface": list(_IN_SFC),
                      "forcings": list(_FORCINGS), "constants": list(_CONSTANTS)},
            "output": {"atmospheric": _IN_ATM + ["vertical_velocity"],
                       "surface": _IN_SFC + ["total_precipitation_6hr"]},
        },
    }
    return to_attr(cfg)


def reduced_config(**model_overrides) -> AttrDict:
    """Small architecture used by tests/smoke (same wiring, ~140 k parameters)."""
    cfg = default_config()
    cfg.model.latent_size = 32
    cfg.model.velocity_vectors = 24
    cfg.model.num_layers = 2
    cfg.model.physblock.velocity_net.hidden_dim = 16
    cfg.model.physblock.reaction.hidden_dim = 48
    cfg.model.physblock.output_proj.hidden_dim = 32
    for k, v in model_overrides.items():
        cfg.model[k] = v
    return cfg


def load_yaml(path: str) -> AttrDict:
    import yaml
    with open(path) as f:
        return to_attr(yaml.safe_load(f))


def feature_layout(cfg) -> SimpleNamespace:
    """Channel bookkeeping the reference's datamodule exposes
    (``data/era5_dataset.py:150-166,262-276``): 176 dynamic inputs, 10 static,
    83 common, 97 outputs for the shipped feature list."""
    levels = list(cfg.features.pressure_levels)
    in_atm = [f"{v}_h{l}" for v in cfg.features.input.atmospheric for l in levels]
    out_atm = [f"{v}_h{l}" for v in cfg.features.output.atmospheric for l in levels]
    in_feats = in_atm + list(cfg.features.input.surface)
    out_feats = out_atm + list(cfg.features.output.surface)
    common = [f for f in out_feats if f in in_feats]
    out_only = [f for f in out_feats if f not in in_feats]
    in_only = [f for f in in_feats if f not in out_feats]
    n_t = cfg.dataset.n_time_inputs
    dyn_single = len(common) + len(in_only)
    n_forc = len(cfg.features.input.forcings) * n_t  # era5_dataset.py:307-309
    return SimpleNamespace(
        num_common_features=len(common),
        num_out_features=len(common) + len(out_only),
        num_in_dyn_features=dyn_single * n_t + n_forc,
        num_in_static_features=len(cfg.features.input.constants),
        output_name_order=common + out_only,
        num_forcings=n_forc,
        dyn_single=dyn_single,
    )


def stub_datamodule(cfg) -> SimpleNamespace:
    """Object exposing the four integers ``Paradis.__init__`` reads from the datamodule."""
    lay = feature_layout(cfg)
    return SimpleNamespace(
        dataset=SimpleNamespace(num_in_dyn_features=lay.num_in_dyn_features,
                                num_in_static_features=lay.num_in_static_features),
        num_common_features=lay.num_common_features,
        num_out_features=lay.num_out_features,
        output_name_order=lay.output_name_order,
    )
