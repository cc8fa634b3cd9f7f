#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the *reference* read-only.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``.  Writes ``tests/golden/*.pt`` and
``tests/golden/default_manifest.json``.  The fixtures are data (inputs, expected
outputs, parameter values); no reference source travels.

What each file pins:
  g1_pad.pt        GeoCyclicPadding index arrays (arange planes) + sha256 for 721x1440
  g2_advect.pt     advection core (projections = identity): fp32 + fp64 outputs and grads
  g3_blocks.pt     CLinear / SepConv / ChannelNorm / GlobalBias / PhysicalDownsample / upsample
  g4_model_*.pt    reduced-config Paradis: state_dict (= seed-42 init), input, output, loss, grads
  g7_amp_*.pt      the same models under torch.autocast(bfloat16) - the reference's shipped bf16-mixed mode
  g5 manifest      default-config state_dict key/shape list (335 entries)
  g6_loss.pt       ParadisLoss weights and values
  c2_rollout.pt    2-step autoregressive rollout (restated trainer loop driving the reference model)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from model.padding import GeoCyclicPadding  # noqa: E402
from model.advection import NeuralSemiLagrangian  # noqa: E402
from model import blocks as rb  # noqa: E402
from model.paradis import Paradis  # noqa: E402
from utils.loss import ParadisLoss  # noqa: E402


class AD(dict):
    """attribute dict with .get(), enough for the reference's cfg accesses"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def to_ad(o):
    if isinstance(o, dict):
        return AD({k: to_ad(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_ad(v) for v in o]
    return o


def load_cfg(**model_over):
    with open(os.path.join(REF, "config/paradis_settings.yaml")) as f:
        raw = yaml.safe_load(f)
    raw["compute"]["compile"] = False
    cfg = to_ad(raw)
    for k, v in model_over.items():
        cur = cfg.model
        parts = k.split("__")
        for p in parts[:-1]:
            cur = cur[p]
        cur[parts[-1]] = v
    return cfg


def stub_dm():
    return types.SimpleNamespace(
        dataset=types.SimpleNamespace(num_in_dyn_features=176, num_in_static_features=10),
        num_common_features=83, num_out_features=97)


def grid(nlat, nlon, poles):
    if poles:
        lat = torch.linspace(-90.0, 90.0, nlat, dtype=torch.float64)
    else:
        d = 180.0 / nlat
        lat = -90.0 + d / 2 + d * torch.arange(nlat, dtype=torch.float64)
    lon = torch.arange(nlon, dtype=torch.float64) * (360.0 / nlon)
    lat_r = torch.deg2rad(lat.to(torch.float32)).to(torch.float32)
    lon_r = torch.deg2rad(lon.to(torch.float32)).to(torch.float32)
    lg, og = torch.meshgrid(lat_r, lon_r, indexing="ij")
    return lat.to(torch.float32), lg.contiguous(), og.contiguous()


def seeded(seed, *shape, scale=1.0, kind="randn"):
    """Inputs are NOT stored: tests regenerate them with the same call (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        return torch.randn(*shape, generator=g) * scale
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def chk(t):
    """cheap checksum stored next to the recipe so a generator mismatch is detected"""
    return [float(t.double().sum()), float(t.double().abs().sum())]


def g1_pad():
    out = {}
    for (H, W) in [(7, 8), (32, 64), (33, 64)]:
        for p in (1, 2, 3):
            x = torch.arange(H * W, dtype=torch.float32).reshape(1, 1, H, W)
            y = GeoCyclicPadding(p)(x)
            out[f"{H}x{W}_p{p}"] = y[0, 0].to(torch.int32)
    x = torch.arange(721 * 1440, dtype=torch.float64).reshape(1, 1, 721, 1440)
    y = GeoCyclicPadding(2)(x)[0, 0].to(torch.int64).numpy()
    out["721x1440_p2_sha256"] = hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()
    torch.save(out, os.path.join(HERE, "g1_pad.pt"))


def g2_advect():
    out = {}
    cfg = load_cfg()
    B = 2
    cases = []
    for (H, W, poles) in [(8, 16, False), (9, 16, True)]:
        for mode in ("bilinear", "bicubic"):
            for scale in (0.1, 1.0, 5.0):
                cases.append((H, W, poles, mode, scale, 6))
    for (H, W, poles) in [(32, 64, False), (33, 64, True)]:
        for mode in ("bilinear", "bicubic"):
            cases.append((H, W, poles, mode, 1.0, 3))
    for n, (H, W, poles, mode, scale, K) in enumerate(cases):
        _, lg, og = grid(H, W, poles)
        adv = NeuralSemiLagrangian(cfg, K, (H, W), K, lg, og, interpolation=mode)
        adv.down_projection = torch.nn.Identity()
        adv.up_projection = torch.nn.Identity()
        seed = 1000 + 10 * n
        f = seeded(seed, B, K, H, W)
        u = seeded(seed + 1, B, K, H, W, scale=scale)
        v = seeded(seed + 2, B, K, H, W, scale=scale)
        ct = seeded(seed + 3, B, K, H, W)
        dt = 0.196887
        rec = {"H": H, "W": W, "poles": poles, "mode": mode, "scale": scale, "K": K, "B": B,
               "seed": seed, "dt": dt, "chk": chk(f) + chk(u) + chk(v) + chk(ct)}
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            a = adv.to(dtype)
            ff, uu, vv = (t.detach().to(dtype).requires_grad_(True) for t in (f, u, v))
            y = a(ff, uu, vv, dt)
            y.backward(ct.to(dtype))
            rec[f"out_{tag}"] = y.detach()
            if tag == "f32":
                rec["gfield_f32"], rec["gu_f32"], rec["gv_f32"] = ff.grad, uu.grad, vv.grad
        out[f"{H}x{W}_{mode}_s{scale}"] = rec
    torch.save(out, os.path.join(HERE, "g2_advect.pt"))


def _fwd_bwd(mod, x, seed):
    x = x.clone().requires_grad_(True)
    y = mod(x)
    torch.manual_seed(seed)
    ct = torch.randn_like(y)
    y.backward(ct)
    rec = {"x": x.detach(), "y": y.detach(), "cot": ct, "gx": x.grad,
           "params": {k: v.detach().clone() for k, v in mod.state_dict().items()},
           "grads": {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}}
    return rec


def g3_blocks():
    out = {}
    torch.manual_seed(7)
    H, W = 12, 16
    out["clinear"] = _fwd_bwd(rb.CLinear(input_dim=10, output_dim=7, mesh_size=(H, W)),
                              torch.randn(2, 10, H, W), 1)
    for k in (5, 7):
        m = rb.SepConv(input_dim=6, output_dim=9, mesh_size=(H, W), kernel_size=k)
        out[f"sepconv_k{k}"] = _fwd_bwd(m, torch.randn(2, 6, H, W), 2)
    m = rb.ChannelNorm(input_dim=20, output_dim=20)
    with torch.no_grad():
        m.weight.normal_(1.0, 0.2)
        m.bias.normal_(0.0, 0.2)
    out["channelnorm"] = _fwd_bwd(m, torch.randn(2, 20, H, W) * 3 + 1, 3)
    for cin, cout, tag in ((8, 8, "noproj"), (8, 24, "proj")):
        m = rb.GlobalBias(input_dim=cin, output_dim=cout, mesh_size=(H, W))
        with torch.no_grad():  # N(0,1e-3) init makes a ~1e-9 map; scale up for a meaningful test
            m.A.normal_(0, 0.3)
            m.U.normal_(0, 0.3)
            m.V.normal_(0, 0.3)
        out[f"globalbias_{tag}"] = _fwd_bwd(m, torch.randn(2, cout, H, W), 4)
    for (h, w) in ((12, 16), (13, 16)):
        for s in (1, 2, 4):
            m = rb.PhysicalDownsample(stride=s)
            out[f"downsample_{h}x{w}_s{s}"] = _fwd_bwd(m, torch.randn(2, 5, h, w), 5)
    for (nlat, nlon, s) in ((12, 16, 1), (12, 16, 2), (13, 16, 2), (33, 64, 4)):
        hc, wc = (nlat - 1) // s + 1, nlon // s
        fake = types.SimpleNamespace(nlat=nlat, nlon=nlon)

        class Up(torch.nn.Module):
            def forward(self, x, fake=fake):
                return Paradis.upsample(fake, x)
        out[f"upsample_{nlat}x{nlon}_s{s}"] = _fwd_bwd(Up(), torch.randn(2, 3, hc, wc), 6)
    # a GMBlock with every ingredient
    torch.manual_seed(11)
    m = rb.GMBlock(layers=["CLinear", "SepConv", "CLinear"], input_dim=10, output_dim=6,
                   mesh_size=(H, W), hidden_dim=12, bias_channels=4, pre_normalize=True,
                   activation_fn=torch.nn.GELU)
    out["gmblock"] = _fwd_bwd(m, torch.randn(2, 10, H, W), 8)
    out["gmblock"]["keys"] = list(m.state_dict().keys())
    torch.save(out, os.path.join(HERE, "g3_blocks.pt"))


REDUCED = dict(latent_size=32, velocity_vectors=24, num_layers=2,
               physblock__velocity_net__hidden_dim=16, physblock__reaction__hidden_dim=48,
               physblock__output_proj__hidden_dim=32)

VARIANTS = {
    "a": dict(nlat=16, nlon=32, poles=False, activation="SiLU", adv_interpolation="bicubic",
              coarsening_factor=1),
    "b": dict(nlat=17, nlon=32, poles=True, activation="GELU", adv_interpolation="bilinear",
              coarsening_factor=2),
    "c": dict(nlat=16, nlon=32, poles=False, activation="SiLU", adv_interpolation="bicubic",
              coarsening_factor=2),
}


def loss_pieces(cfg, lat_deg):
    levels = list(cfg.features.pressure_levels)
    in_atm = [f"{v}_h{l}" for v in cfg.features.input.atmospheric for l in levels]
    out_atm = [f"{v}_h{l}" for v in cfg.features.output.atmospheric for l in levels]
    in_feats = in_atm + list(cfg.features.input.surface)
    out_feats = out_atm + list(cfg.features.output.surface)
    common = [f for f in out_feats if f in in_feats]
    order = common + [f for f in out_feats if f not in in_feats]
    import re
    w = torch.zeros(len(order))
    for i, f in enumerate(order):
        base = re.sub(r"_h\d+$", "", f)
        vw = cfg.training.variable_loss_weights
        w[i] = vw.atmospheric[base] if base in vw.atmospheric else vw.surface[base]
    return ParadisLoss(loss_function=cfg.training.loss_function.type, lat_grid=lat_deg,
                       pressure_levels=torch.tensor(levels, dtype=torch.float32),
                       num_features=len(order), num_surface_vars=len(cfg.features.output.surface),
                       var_loss_weights=w, output_name_order=order,
                       delta_loss=cfg.training.loss_function.delta_loss,
                       apply_latitude_weights=cfg.training.loss_function.lat_weights), w, order


def build_model(variant, seed=42):
    v = VARIANTS[variant]
    over = dict(REDUCED)
    over.update(activation=v["activation"], adv_interpolation=v["adv_interpolation"],
                coarsening_factor=v["coarsening_factor"])
    cfg = load_cfg(**over)
    lat_deg, lg, og = grid(v["nlat"], v["nlon"], v["poles"])
    torch.manual_seed(seed)
    model = Paradis(stub_dm(), cfg, lg, og)
    return cfg, model, lat_deg, lg, og


def g4_models():
    for variant in VARIANTS:
        cfg, model, lat_deg, lg, og = build_model(variant)
        init_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # perturb the zero/constant-initialised parameters so every path carries signal
        torch.manual_seed(4242)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith((".A", ".U", ".V")):
                    p.normal_(0, 0.2)
                elif n.endswith("ChannelNorm.bias") or (n.endswith(".bias") and p.dim() == 1):
                    p.normal_(0, 0.1)
                elif n == "alpha_adv":
                    p.normal_(-1.0, 0.5)
        loss_fn, _, _ = loss_pieces(cfg, lat_deg)
        B = 2
        x = seeded(1234, B, 186, lg.shape[0], lg.shape[1])
        x[:, -2] = lg
        x[:, -1] = og
        tgt = seeded(1235, B, 97, lg.shape[0], lg.shape[1])
        x.requires_grad_(True)
        y = model(x)
        loss = loss_fn(y, tgt)
        loss.backward()
        rec = {"variant": VARIANTS[variant], "reduced": REDUCED, "init_state": init_sd,
               "state": {k: v.detach().clone() for k, v in model.state_dict().items()},
               "x_seed": 1234, "target_seed": 1235, "B": B, "chk": chk(x.detach()) + chk(tgt),
               "y": y.detach(), "loss": loss.detach(),
               "gx_sub": x.grad[:, ::9].clone(), "lat_deg": lat_deg, "lat_grid": lg, "lon_grid": og,
               "grads": {k: p.grad.clone() for k, p in model.named_parameters()}}
        torch.save(rec, os.path.join(HERE, f"g4_model_{variant}.pt"))


def g7_amp():
    """The reference in its SHIPPED training mode: ``use_amp: true`` (config/paradis_settings.yaml:75) ->
    ``precision="bf16-mixed"`` (train.py:56), i.e. forward and loss under ``torch.autocast(dtype=torch.bfloat16)``
    (here on the CPU: the same autocast op lists route every conv2d / linear / matmul through bf16 and keep
    grid_sampler in fp32).  Same reduced models, states and inputs as g4 (the states are read back from the g4 files);
    stored: output, loss, input-gradient subsample and every parameter gradient - the pins of the PARADIS_GEMM_BF16
    path (tests/test_hip_amp.py) at a bf16-level tolerance."""
    for variant in ("a", "b"):
        cfg, model, lat_deg, lg, og = build_model(variant)
        g4 = torch.load(os.path.join(HERE, f"g4_model_{variant}.pt"), weights_only=False)
        model.load_state_dict(g4["state"])
        loss_fn, _, _ = loss_pieces(cfg, lat_deg)
        B = 2
        x = seeded(1234, B, 186, lg.shape[0], lg.shape[1])
        x[:, -2] = lg
        x[:, -1] = og
        tgt = seeded(1235, B, 97, lg.shape[0], lg.shape[1])
        x.requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = model(x)
            loss = loss_fn(y, tgt)
        loss.backward()
        rec = {"variant": VARIANTS[variant], "chk": chk(x.detach()) + chk(tgt), "y_dtype": str(y.dtype),
               "y": y.detach().float(), "loss": loss.detach().float(), "gx_sub": x.grad[:, ::9].clone(),
               "grads": {k: p.grad.clone() for k, p in model.named_parameters()}}
        torch.save(rec, os.path.join(HERE, f"g7_amp_{variant}.pt"))


def g5_manifest():
    cfg = load_cfg()
    _, lg, og = grid(32, 64, False)
    torch.manual_seed(42)
    model = Paradis(stub_dm(), cfg, lg, og)
    man = {"num_parameters": sum(p.numel() for p in model.parameters()),
           "entries": [[k, list(v.shape)] for k, v in model.state_dict().items()],
           "dt": model.dt,
           # a few cheap statistics of the seed-42 default init (init parity check)
           "init_probe": {k: [float(v.double().sum()), float(v.double().abs().max())]
                          for k, v in list(model.state_dict().items())[:12]}}
    with open(os.path.join(HERE, "default_manifest.json"), "w") as f:
        json.dump(man, f, indent=0)


def g6_loss():
    cfg = load_cfg()
    out = {}
    for (nlat, nlon, poles) in ((32, 64, False), (33, 64, True), (721, 1440, True)):
        lat_deg, _, _ = grid(nlat, nlon, poles)
        for kind in ("reversed_huber", "mse"):
            cfg.training.loss_function.type = kind
            fn, vw, order = loss_pieces(cfg, lat_deg)
            rec = {"lat_deg": lat_deg, "lat_weights": fn.lat_weights, "feature_weights": fn.feature_weights,
                   "var_weights": vw, "order": order}
            if nlat < 100:
                p = seeded(99, 2, 97, nlat, nlon, scale=1.5).requires_grad_(True)
                t = seeded(100, 2, 97, nlat, nlon)
                l = fn(p, t)
                l.backward()
                rec.update(pred_seed=99, target_seed=100, chk=chk(p.detach()) + chk(t),
                           loss=l.detach(), gpred_sub=p.grad[:, ::8, ::2, ::4].clone())
            out[f"{nlat}x{nlon}_{kind}"] = rec
    torch.save(out, os.path.join(HERE, "g6_loss.pt"))


def c2_rollout():
    """trainer.py:498-587 restated (lightning is not importable) driving the reference model+loss."""
    cfg, model, lat_deg, lg, og = build_model("a")
    loss_fn, _, _ = loss_pieces(cfg, lat_deg)
    H, W = lg.shape
    B, S, ncom = 2, 2, 83
    inp = seeded(77, B, 1, 166, H, W)
    tgt = seeded(78, B, S, 97, H, W)
    forc = seeded(79, B, S, H, W, 10, kind="rand")
    const = seeded(80, B, 1, H, W, 10)
    constants = const[:, :1].permute(0, 1, 4, 2, 3)
    forcings = forc.permute(0, 1, 4, 2, 3)
    cur = inp
    chunk = 0.0
    outs = []
    for s in range(S):
        mi = torch.cat([cur, forcings[:, s].unsqueeze(1), constants], dim=2).squeeze(1)
        y = model(mi)
        outs.append(y.detach())
        chunk = chunk + loss_fn(y, tgt[:, s]) / S
        cur = torch.cat([mi[:, ncom:2 * ncom], y[:, :ncom]], dim=1).unsqueeze(1)
    chunk.backward()
    rec = {"state": {k: v.detach().clone() for k, v in model.state_dict().items()},
           "seeds": [77, 78, 79, 80], "B": B, "S": S,
           "chk": chk(inp) + chk(tgt) + chk(forc) + chk(const),
           "outputs": outs, "loss": chunk.detach(), "lat_deg": lat_deg, "lat_grid": lg, "lon_grid": og,
           "grad_norms": {k: float(p.grad.norm()) for k, p in model.named_parameters()},
           "grad_alpha": model.alpha_adv.grad.clone()}
    torch.save(rec, os.path.join(HERE, "c2_rollout.pt"))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "c2", "g7"]
    fns = {"g1": g1_pad, "g2": g2_advect, "g3": g3_blocks, "g4": g4_models, "g5": g5_manifest,
           "g6": g6_loss, "c2": c2_rollout, "g7": g7_amp}
    for w in which:
        fns[w]()
        print("wrote", w)
