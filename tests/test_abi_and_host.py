"""CPU-only checks: the C-ABI library loads and exports every symbol include/paradis_hip.h
declares (no compute calls), argument validation that needs no GPU, the drop-in module surface
(names, state-dict keys, init parity with the reference) and the host harness."""
import ctypes
import json
import os
import re

import pytest
import torch

from paradis_model_amd.config import (default_config, feature_layout, reduced_config, stub_datamodule)
from tests._util import GOLDEN, load_golden, make_grid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "paradis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(paradis_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from paradis_model_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/paradis_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= set(names)
    assert _lib.lib.paradis_abi_version() == 9


def test_argument_validation_without_gpu():
    """Rejected arguments return before any HIP call, so these run on a CPU-only box."""
    from paradis_model_amd import _lib
    L = _lib.lib
    assert L.paradis_geocyclic_pad_fwd(None, None, 1, 8, 7, 1, None) == 1      # odd longitude count
    assert "even" in _lib.last_error()
    assert L.paradis_geocyclic_pad_fwd(None, None, 1, 4, 8, 3, None) == 1      # pad > H-2
    assert L.paradis_dwconv_geo_fwd(None, None, None, None, 1, 4, 16, 32, 4, None) == 1   # even kernel
    assert L.paradis_sl_advect_fwd(None, None, None, None, None, None, None, None, 1, 1, 16, 32, 0, 0, 0,
                                   0.1, 0.0, 0.0, 1.0, 1.0, 3, 0, None, None) == 1              # bad mode
    assert L.paradis_pw_gemm_fwd(None, None, None, 0, None, None, None, None, None, None, 0, None, None, None, 1, 0, 4, 4, 0, 0, 0, 0, None) == 1
    assert L.paradis_avgpool_geo_fwd(None, None, 1, 16, 32, 0, None) == 1      # stride < 1
    # data feed (row f4): window longer than the series; unknown forcing code; no variables
    import ctypes
    codes = (ctypes.c_int * 2)(0, 9)
    assert L.paradis_forcings(None, None, None, 0, 1, 1, 8, 16, 2, codes, 2, 0.0, 1.0, None, None, None) == 1
    assert L.paradis_forcings(None, None, None, 0, 1, 3, 8, 16, 2, codes, 2, 0.0, 1.0, None, None, None) == 1
    assert "workspace" in _lib.last_error() or "code" in _lib.last_error()
    assert L.paradis_forcings(None, None, None, 0, 1, 3, 8, 16, 2, codes, 0, 0.0, 1.0, None, None, None) == 1
    assert L.paradis_forcings_ws_bytes(2, 4) >= 8 * 15 * 8 + 8 * 49 * 4
    assert L.paradis_normalize_features(None, None, None, None, 4, 0, 1e-12, 0, None) == 1
    assert L.paradis_normalize_features(None, None, None, None, 0, 5, 1e-12, 0, None) == 0
    # zero-sized batches are accepted and do nothing
    assert L.paradis_geocyclic_pad_fwd(None, None, 0, 8, 8, 1, None) == 0
    assert L.paradis_sl_advect_ws_bytes(2, 3, 8, 16, 0) >= 2 * 3 * 4 * 4


def test_ops_refuse_cpu_tensors():
    from paradis_model_amd import feed, ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        feed.normalize_features_(torch.zeros(2, 3), [0, 0, 0], [0.0] * 3, [1.0] * 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pointwise(torch.randn(1, 4, 8, 8), torch.randn(3, 4, 1, 1))
    with pytest.raises(RuntimeError):
        ops.channel_norm(torch.randn(1, 4, 8, 8), torch.ones(4), torch.zeros(4))


def test_module_surface_matches_reference_names():
    import paradis_model_amd.model as M
    for name in ("Paradis", "NeuralSemiLagrangian", "SemiLagrangianAdvection", "GeoCyclicPadding",
                 "GeocyclicPadding", "GMBlock", "PhysicalDownsample", "SepConv", "CLinear", "ChannelNorm",
                 "GlobalBias", "BLOCK_REGISTRY", "init_module_convs", "init_conv2d_default",
                 "get_scaled_timestep"):
        assert hasattr(M, name)
    assert M.SemiLagrangianAdvection is M.NeuralSemiLagrangian
    assert set(M.BLOCK_REGISTRY) == {"SepConv", "CLinear", "ChannelNorm", "GlobalBias"}
    assert abs(M.get_scaled_timestep(21600) - 21600 * 7.29212e-5) < 1e-12


def test_default_state_dict_manifest_and_init_probe():
    from paradis_model_amd.model import Paradis
    with open(os.path.join(GOLDEN, "default_manifest.json")) as f:
        man = json.load(f)
    cfg = default_config()
    _, lg, og = make_grid(32, 64, False)
    torch.manual_seed(42)
    m = Paradis(stub_datamodule(cfg), cfg, lg, og)
    sd = m.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == man["entries"]
    assert sum(p.numel() for p in m.parameters()) == man["num_parameters"] == 60038475
    assert abs(m.dt - man["dt"]) < 1e-15
    for k, (s, a) in man["init_probe"].items():      # same seed -> same initial weights as the reference
        assert abs(float(sd[k].double().sum()) - s) <= 1e-9 * max(1.0, abs(s)), k
    # weight holders stay nn.Conv2d / nn.Linear (optimiser grouping in the reference trainer)
    assert isinstance(m.reaction[0][1].conv, torch.nn.Conv2d)
    assert isinstance(m.velocity_nets[0][2].projection, torch.nn.Linear)
    assert "advection.0.lat_grid" not in sd and hasattr(m.advection[0], "lat_grid")


@pytest.mark.parametrize("variant", ["a", "b", "c"])
def test_reduced_init_is_bit_identical_to_reference(variant):
    from paradis_model_amd.model import Paradis
    rec = load_golden(f"g4_model_{variant}.pt")
    v = rec["variant"]
    cfg = reduced_config(activation=v["activation"], adv_interpolation=v["adv_interpolation"],
                         coarsening_factor=v["coarsening_factor"])
    torch.manual_seed(42)
    m = Paradis(stub_datamodule(cfg), cfg, rec["lat_grid"], rec["lon_grid"])
    sd = m.state_dict()
    assert list(sd) == list(rec["init_state"])
    for k in sd:
        assert torch.equal(sd[k], rec["init_state"][k]), k
    m.load_state_dict(rec["state"], strict=True)        # trained-state fixtures load strictly


def test_config_errors_match_reference_behaviour():
    from paradis_model_amd.model import GMBlock, Paradis
    cfg = reduced_config(activation="ReLU")
    _, lg, og = make_grid(16, 32, False)
    with pytest.raises(ValueError, match="Unknown activation_fn"):
        Paradis(stub_datamodule(cfg), cfg, lg, og)
    cfg = reduced_config(coarsening_factor=0)
    with pytest.raises(ValueError, match="Coarsening factor"):
        Paradis(stub_datamodule(cfg), cfg, lg, og)
    with pytest.raises(ValueError, match="at least one layer"):
        GMBlock(layers=[], input_dim=4, output_dim=4, mesh_size=(8, 16))
    with pytest.raises(ValueError, match="Unknown layer type"):
        GMBlock(layers=["Conv3"], input_dim=4, output_dim=4, mesh_size=(8, 16))


def test_feature_layout_and_loss_weights():
    from paradis_model_amd.loss import build_loss
    cfg = default_config()
    lay = feature_layout(cfg)
    assert (lay.num_in_dyn_features, lay.num_in_static_features, lay.num_common_features,
            lay.num_out_features) == (176, 10, 83, 97)
    g = load_golden("g6_loss.pt")
    for key, rec in g.items():
        kind = key.split("_", 1)[1]
        cfg.training.loss_function.type = kind
        fn = build_loss(cfg, rec["lat_deg"])
        assert torch.equal(fn.feature_weights, rec["feature_weights"])
        assert torch.allclose(fn.lat_weights, rec["lat_weights"], rtol=1e-6, atol=0)
        assert fn.output_name_order == rec["order"]


def test_synthetic_batch_shapes_and_rollout_glue():
    from paradis_model_amd.harness import assemble_model_input, next_input, synthetic_batch
    inp, tgt, forc, const = synthetic_batch(32, 64, False, 3, 2)
    assert inp.shape == (3, 1, 166, 32, 64) and tgt.shape == (3, 2, 97, 32, 64)
    assert forc.shape == (3, 2, 32, 64, 10) and const.shape == (3, 1, 32, 64, 10)
    mi = assemble_model_input(inp, forc.permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1),
                              const[:, :1].permute(0, 1, 4, 2, 3))
    assert mi.shape == (3, 186, 32, 64)
    out = torch.randn(3, 97, 32, 64)
    nxt = next_input(mi, out, 83, 2)
    assert nxt.shape == (3, 166, 32, 64)
    assert torch.equal(nxt[:, :83], mi[:, 83:166]) and torch.equal(nxt[:, 83:], out[:, :83])
    assert torch.equal(next_input(mi, out, 83, 1), out[:, :83])


def test_gemm_scheme_selection(monkeypatch):
    """Host logic of the GEMM arithmetic switch (PARADIS_GEMM): the default is the reference-width bf16x3
    decomposition; f16x2 is opt-in; the ops carry the scheme as an explicit integer argument."""
    from paradis_model_amd import ops
    for name, code in (("f16x2", 2), ("bf16x3", 3), ("split", 3), ("exact", 0)):
        monkeypatch.setenv("PARADIS_GEMM", name)
        assert ops._scheme_from_env() == code
    monkeypatch.setenv("PARADIS_GEMM", "fp8")
    with pytest.raises(ValueError):
        ops._scheme_from_env()
    monkeypatch.delenv("PARADIS_GEMM")
    assert ops._scheme_from_env() == ops.GEMM_BF16X3                      # the default
    for op in ("pointwise", "pw_gemm_dgrad", "pw_gemm_wgrad"):
        schema = str(ops.OPS[op]._schema)
        assert "int scheme" in schema, schema
    assert not hasattr(ops, "TRACED") and not hasattr(ops, "_amax_attach")   # no hidden per-tensor / global state


def test_weight_image_cache_invalidation():
    """Cached split weight images are dropped by any optimiser step (global post-hook), by the version counter
    and by ``ops.weights_updated()``; inference tensors have no version counter and must not raise."""
    import torch
    from paradis_model_amd import ops
    ops._ensure_step_hook()       # (registered when the first weight image is cached - not at import: round 5)
    ops._ensure_step_hook()       # idempotent
    e0 = ops.WEIGHT_EPOCH
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    torch.optim.SGD([p], lr=0.1).step()                                   # a foreign optimiser
    assert ops.WEIGHT_EPOCH == e0 + 1
    e1 = ops.WEIGHT_EPOCH
    ops.weights_updated()
    assert ops.WEIGHT_EPOCH == e1 + 1
    with torch.inference_mode():
        t = torch.ones(2)
    assert ops._version_of(t) == 0
    q = torch.ones(2)
    v = ops._version_of(q)
    q.add_(1)
    assert ops._version_of(q) == v + 1


def test_weight_gradient_slab_count_is_even_or_one():
    """The weight gradient cancels the offset between K-range slabs of alternating sign, which needs an even number of
    them.  The host picks the slab count from the tile count (768 resident workgroups / tiles): 1536 x 384 would get 21.
    It is rounded down to an even number (csrc/gemm.hip wgrad_splits); one slab (weight matrices beyond 384 tiles, none in
    this model) has no partner and keeps the offset - documented, not cancelled."""
    from paradis_model_amd._lib import lib
    shapes = [(1024, 186), (384, 1024), (1536, 384), (768, 1024), (1024, 768), (1024, 1024), (896, 1152), (896, 896),
              (1024, 896), (768, 768), (97, 768), (2048, 1536), (128, 128), (640, 640)]
    for (Co, Ci) in shapes:
        for (B, N) in ((4, 2048), (1, 32768), (32, 2048), (1, 16)):
            S = lib.paradis_pw_gemm_wgrad_slabs(B, Co, Ci, N)
            assert S >= 1 and (S == 1 or S % 2 == 0), ((Co, Ci, B, N), S)
    assert lib.paradis_pw_gemm_wgrad_slabs(4, 1536, 384, 2048) == 20      # 768 / 36 tiles = 21 -> 20
