"""The custom-op boundary on the GPU: ``torch.compile(fullgraph=True)`` of the drop-in model (what the
reference trainer does with ``compute.compile``, ``trainer.py:261-269``), per-module ``_compile()``
(``model/paradis.py:195-206``), autocast, ``torch.library.opcheck`` of the op registrations."""
import copy

import pytest
import torch

from paradis_model_amd.config import reduced_config, stub_datamodule
from tests._util import make_grid, max_rel, seeded

pytestmark = pytest.mark.gpu


def _build(cfg, H=16, W=32):
    from paradis_model_amd.model import Paradis
    _, lg, og = make_grid(H, W, False)
    torch.manual_seed(42)
    m = Paradis(stub_datamodule(cfg), cfg, lg, og)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for n, p in m.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    return m.cuda()


def _fwd_bwd(model, x):
    model.zero_grad(set_to_none=True)
    xd = x.clone().requires_grad_(True)
    y = model(xd)
    y.square().mean().backward()
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    return y.detach(), xd.grad.detach(), grads


@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_compile_fullgraph_matches_eager(backend):
    """reference trainer.py:262-267: model.compile(mode="default", fullgraph=True, dynamic=False, backend=...)"""
    cfg = reduced_config()
    eager = _build(cfg)
    comp = copy.deepcopy(eager)
    comp.compile(mode="default", fullgraph=True, dynamic=False, backend=backend)
    x = seeded(1, 2, 186, 16, 32).cuda()
    y0, gx0, g0 = _fwd_bwd(eager, x)
    y1, gx1, g1 = _fwd_bwd(comp, x)
    assert max_rel(y1, y0) <= 1e-6
    assert max_rel(gx1, gx0) <= 1e-5     # float atomics in a few reductions: not bit-reproducible
    assert max_rel(g1, g0) <= 1e-5
    # second call reuses the compiled graph
    y2, _, _ = _fwd_bwd(comp, x)
    assert max_rel(y2, y0) <= 1e-6


def test_compile_under_bf16_autocast_matches_the_eager_bf16_mixed_step():
    """The reference's two switches together (compute.compile + use_amp, trainer.py:261-269 with train.py:56): the traced
    graph under torch.autocast(bfloat16) runs the bf16-mixed scheme with fp32-stored activations (a traced graph makes no
    storage requests) and the bf16 d(pre-activation) of ``act_backward`` - whose fake kernel must name the dtype the HIP
    kernel returns.  Against the eager bf16-mixed step (bf16 storage on): same values up to rounding flips."""
    from tests._util import rms_rel
    cfg = reduced_config()
    eager = _build(cfg)
    comp = copy.deepcopy(eager)
    comp.compile(mode="default", fullgraph=True, dynamic=False, backend="aot_eager")
    x = seeded(5, 2, 186, 16, 32).cuda()

    def run(model):
        model.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(xd)
            loss = y.square().mean()
        loss.backward()
        return y.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}

    y0, g0 = run(eager)
    y1, g1 = run(comp)
    assert y1.dtype == torch.float32 and rms_rel(y1, y0) <= 2e-3
    errs = sorted(rms_rel(g1[n], g0[n]) for n in g0 if float(g0[n].abs().max()) > 0)
    assert errs[len(errs) // 2] <= 2e-3 and errs[-1] <= 3e-2, (errs[len(errs) // 2], errs[-1])


def test_per_module_compile_matches_eager():
    """reference model/paradis.py:195-206 (compute.compile == "modules")"""
    cfg = reduced_config()
    eager = _build(cfg)
    comp = copy.deepcopy(eager)
    comp._compile()
    x = seeded(2, 2, 186, 16, 32).cuda()
    y0, gx0, g0 = _fwd_bwd(eager, x)
    y1, gx1, g1 = _fwd_bwd(comp, x)
    assert max_rel(y1, y0) <= 1e-6 and max_rel(gx1, gx0) <= 1e-5 and max_rel(g1, g0) <= 1e-5
    # the trainer strips "._orig_mod." when it loads such a checkpoint (reference trainer.py:222-258)
    keys = {k.replace("._orig_mod", "") for k in comp.state_dict()}
    assert keys == set(eager.state_dict())


def test_autocast_modes():
    """precision="bf16-mixed" (reference train.py:56, its shipped default).  Rounds 1-4: every op widened its inputs to
    fp32 and the result equalled the fp32 run.  Round 5: the pointwise GEMMs - autocast's conv2d - run the one-product
    bf16 scheme inside bfloat16 autocast (tests/test_hip_amp.py pins it); every other op still widens (never narrower than
    the reference, which keeps grid_sampler in fp32 too), outputs stay fp32 tensors, a bf16 input is widened, not
    rejected; under FLOAT16 autocast nothing changes: the fp32 result bit for bit."""
    from paradis_model_amd import ops
    cfg = reduced_config()
    model = _build(cfg)
    x = seeded(3, 1, 186, 16, 32).cuda()
    with torch.no_grad():
        y0 = model(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y1 = model(x)
            y2 = model(x.to(torch.bfloat16))
            f = seeded(4, 1, 6, 16, 32).cuda()
            w = seeded(5, 6, 1, 5, 5).cuda()
            with torch.autocast("cuda", enabled=False):
                ref = ops.dwconv_geo(f, w)
            assert torch.equal(ops.dwconv_geo(f, w), ref)              # a non-GEMM op: fp32 inside bf16 autocast
        with torch.autocast("cuda", dtype=torch.float16):
            y3 = model(x)
    assert y1.dtype == torch.float32 and 1e-4 < max_rel(y1, y0) < 0.05     # the bf16 path ran
    assert y2.dtype == torch.float32 and max_rel(y2, y0) < 0.05
    assert y3.dtype == torch.float32 and torch.equal(y3, y0)


def test_non_fp32_and_cpu_tensors_raise():
    from paradis_model_amd import ops
    x = torch.randn(1, 2, 8, 16)
    with pytest.raises(RuntimeError):
        ops.geocyclic_pad(x, 1)
    with pytest.raises(RuntimeError):
        ops.geocyclic_pad(x.cuda().double(), 1)
    with pytest.raises(RuntimeError):
        torch.ops.paradis.add(x.cuda().double(), x.cuda().double())


def test_opcheck_registrations():
    """torch.library.opcheck: schema, fake kernel vs real kernel (shapes/strides/dtypes), autograd
    registration and AOT dispatch of representative ops."""
    from paradis_model_amd import ops
    from torch.library import opcheck
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    H, W = 8, 16
    _, lg, og = make_grid(H, W, False)
    geom = ops.AdvectGeometry(lg, og)
    sl, cl, lc, lo = geom.tables(torch.device("cuda"), 2)
    g = lambda *s: torch.randn(*s, device="cuda", requires_grad=True)
    opcheck(torch.ops.paradis.geocyclic_pad.default, (g(1, 2, H, W), 2), test_utils=tests)
    opcheck(torch.ops.paradis.sl_advect.default,
            (g(1, 2, H, W), g(1, 2, H, W), g(1, 2, H, W), sl, cl, lc, lo, 0.2, geom.min_lat, geom.min_lon,
             geom.d_lat, geom.d_lon, 2, 0), test_utils=tests)
    opcheck(torch.ops.paradis.dwconv_geo.default, (g(1, 3, H, W), g(3, 1, 5, 5), None), test_utils=tests)
    opcheck(torch.ops.paradis.channel_norm.default, (g(1, 4, H, W), g(1, 2, H, W), g(6), g(6), 1e-5),
            test_utils=tests)
    opcheck(torch.ops.paradis.pointwise.default,
            (g(1, 4, H, W), g(5, 4, 1, 1), g(5), None, g(1, 5, H, W), 1, None, 0, False, None, None, True, ops.GEMM_BF16X3,
             None), test_utils=tests)
    opcheck(torch.ops.paradis.pointwise.default,      # gated epilogue: blend with the residual
            (g(1, 4, H, W), g(5, 4, 1, 1), g(5), None, g(1, 5, H, W), 1, None, 0, False, None, None, True, ops.GEMM_BF16X3,
             g(5)), test_utils=tests)
    opcheck(torch.ops.paradis.dwconv_geo_bwd.default, (g(1, 3, H, W), g(1, 3, H, W), g(3, 1, 5, 5), g(1, 3, H, W), True),
            test_utils=("test_schema", "test_faketensor"))
    opcheck(torch.ops.paradis.gated_blend.default, (g(1, 3, H, W), g(1, 3, H, W), g(3)), test_utils=tests)
    opcheck(torch.ops.paradis.concat_channels.default, ([g(1, 3, H, W), g(1, 2, H, W)],), test_utils=tests)


def test_weight_images_follow_every_kind_of_weight_update():
    """Verdict r5 item 9: no write to a parameter may be missed.  The split weight images are rebuilt by every forward
    call, so a torch in-place update, a write through ``.data`` (no version-counter bump), a raw-pointer write by the
    HIP optimiser and a new tensor at the same address are all seen WITHOUT any hint; inside the opt-in
    ``frozen_weights()`` block the cache follows version counter / optimiser steps / ``weights_updated()``."""
    from paradis_model_amd import ops
    if ops.GEMM_SCHEME == ops.GEMM_EXACT:
        pytest.skip("exact f32 GEMMs keep no weight images")
    x = torch.randn(2, 32, 8, 16, device="cuda")
    w = torch.nn.Parameter(torch.randn(48, 32, 1, 1, device="cuda"))
    ref = lambda: torch.einsum("oc,bchw->bohw", w.detach().reshape(48, 32).double(), x.double()).float()
    with torch.no_grad():
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
        assert not ops._IMAGES                               # nothing cached
        w.mul_(2.0)                                          # torch in-place update
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
        v = w._version
        w.data.mul_(2.0)                                     # no version bump, no hint
        assert w._version == v
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
        import ctypes
        from paradis_model_amd._lib import lib
        half = torch.full((1,), 0.5, device="cuda")
        assert lib.paradis_scale(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(half.data_ptr()),
                                 ctypes.c_void_p(w.data_ptr()), w.numel(), ops.stream_ptr()) == 0   # raw-pointer kernel
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
    from paradis_model_amd.optim import AdamW
    opt = AdamW([w], lr=0.1)
    # the backward uses the W^T image its own forward wrote; a .data write AFTER the step is seen by the next forward
    xg = x.clone().requires_grad_(True)
    ops.pointwise(xg, w).square().mean().backward()
    gref = torch.autograd.grad((torch.einsum("oc,bchw->bohw", w.detach().reshape(48, 32), xg)).square().mean(), xg)[0]
    assert max_rel(xg.grad, gref) < 1e-5
    opt.step()                                               # HIP kernel writes through raw pointers
    w.data.add_(0.25)
    with torch.no_grad():
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
    # opt-in cache
    with ops.frozen_weights(), torch.no_grad():
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
        n_img = len(ops._IMAGES)
        assert n_img >= 1
        ops.pointwise(x, w)
        assert len(ops._IMAGES) == n_img                     # cached
        w.mul_(2.0)                                          # version counter
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
        w.data.mul_(0.5)
        ops.weights_updated()                                # the declared-frozen mode needs the hint
        assert max_rel(ops.pointwise(x, w), ref()) < 1e-5
    assert not ops._IMAGES
