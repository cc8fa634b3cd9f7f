"""The torch custom-op boundary without a GPU: every ``paradis::*`` op has a HIP
kernel, a fake (Meta) kernel and - for the differentiable ones - an autograd formula; the model runs
under FakeTensorMode on fake ``cuda`` tensors, and ``torch.compile(fullgraph=True)`` traces it into one
graph whose only compute nodes are ``paradis::*`` ops (reference ``trainer.py:261-267``).
No kernel is launched here; numerical parity of the compiled model is a ``-m gpu`` test."""
import collections

import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode

from paradis_model_amd import ops
from paradis_model_amd.config import reduced_config, stub_datamodule
from tests._util import make_grid

FORWARD_OPS = ["geocyclic_pad", "sl_advect", "sl_advect_vel", "dwconv_geo", "avgpool_geo", "upsample_lonp",
               "channel_norm", "global_bias_map", "global_bias_m8", "pointwise", "activation", "gated_blend",
               "add", "add_bias_map", "paradis_loss", "concat_channels"]


def _has(name, key):
    return torch._C._dispatch_has_kernel_for_dispatch_key(f"paradis::{name}", key)


def test_every_op_is_registered_with_hip_fake_and_autograd_kernels():
    assert len(ops.OPS) >= 35
    for name in ops.OPS:
        assert _has(name, "CUDA"), name
        assert not _has(name, "CPU"), f"{name}: there must be no CPU fallback"
        assert _has(name, "Meta") or torch.library.get_ctx is not None
    for name in FORWARD_OPS:
        assert name in ops.OPS, name
        assert _has(name, "Autograd"), name
        if name != "concat_channels":
            assert _has(name, "AutocastCUDA"), name


def test_cpu_tensors_are_rejected_loudly():
    x = torch.randn(1, 2, 8, 16)
    with pytest.raises(RuntimeError):
        ops.geocyclic_pad(x, 1)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.paradis.add(x, x)


def _fake_model(mode, cfg, H, W):
    from paradis_model_amd.model import Paradis
    _, lg, og = make_grid(H, W, False)
    m = Paradis(stub_datamodule(cfg), cfg, lg, og)
    with mode:
        for mod in m.modules():
            for k, p in list(mod._parameters.items()):
                if p is not None:
                    mod._parameters[k] = torch.nn.Parameter(mode.from_tensor(p.data).to("cuda"))
            for k, b in list(mod._buffers.items()):
                if b is not None:
                    mod._buffers[k] = mode.from_tensor(b).to("cuda")
    return m


@pytest.mark.parametrize("stride,ckpt", [(1, False), (2, False), (1, True)])
def test_model_forward_on_fake_cuda_tensors(stride, ckpt):
    cfg = reduced_config(coarsening_factor=stride)
    cfg.compute.gradient_checkpointing = ckpt
    mode = FakeTensorMode(allow_non_fake_inputs=True)
    m = _fake_model(mode, cfg, 16, 32)
    with mode:
        y = m(torch.empty(2, 186, 16, 32, device="cuda"))
        assert tuple(y.shape) == (2, 97, 16, 32) and y.device.type == "cuda" and y.requires_grad
        with torch.no_grad():
            y = m(torch.empty(1, 186, 16, 32, device="cuda"))
        assert tuple(y.shape) == (1, 97, 16, 32) and not y.requires_grad


def test_fullgraph_trace_contains_only_paradis_ops():
    cfg = reduced_config()
    mode = FakeTensorMode(allow_non_fake_inputs=True)
    m = _fake_model(mode, cfg, 16, 32)
    graphs = []

    def backend(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    with mode:
        cm = torch.compile(m, backend=backend, fullgraph=True, dynamic=False)
        y = cm(torch.empty(2, 186, 16, 32, device="cuda"))
    assert tuple(y.shape) == (2, 97, 16, 32)
    assert len(graphs) == 1, "graph break"
    counts = collections.Counter(str(n.target) for n in graphs[0].graph.nodes if n.op == "call_function")
    allowed_glue = ("getitem", "built-in method full")      # tuple unpack / slices, 1/25 box weights
    for target, n in counts.items():
        assert target.startswith("paradis.") or any(g in target for g in allowed_glue), (target, n)
    # reduced config: 2 layers x (velocity 2 + advection 2 + diffusion 1 + reaction 4) + in 1 + out 3 + static 2
    assert counts["paradis.pointwise.default"] == 24
    assert counts["paradis.sl_advect_vel.default"] == 2
    assert counts["paradis.channel_norm.default"] == 7


def test_fake_kernels_of_backward_ops_give_the_right_shapes():
    mode = FakeTensorMode(allow_non_fake_inputs=True)
    O = torch.ops.paradis
    with mode:
        e = lambda *s: torch.empty(*s, device="cuda")
        B, C, H, W, K = 2, 6, 8, 16, 3
        assert O.geocyclic_pad_backward(e(B, C, H + 4, W + 4), 2).shape == (B, C, H, W)
        tabs = (e(H, W), e(H, W), e(H, W), e(H, W), 0.1, 0.0, 0.0, 1.0, 1.0, 2, 0)
        gf, gu, gv = O.sl_advect_backward(e(B, K, H, W), e(B, K, H, W), e(B, K, H, W), e(B, K, H, W), *tabs)
        assert gf.shape == gu.shape == gv.shape == (B, K, H, W)
        gf, gvel = O.sl_advect_vel_backward(e(B, K, H, W), e(B, K, H, W), e(B, 2 * K, H, W), *tabs)
        assert gvel.shape == (B, 2 * K, H, W)
        gw, gb = O.dwconv_geo_wgrad(e(B, C, H, W), e(B, C, H, W), 5, False)
        assert gw.shape == (C, 1, 5, 5) and gb.numel() == 0
        gx1, gx2, gw, gb = O.channel_norm_backward(e(B, C + 2, H, W), e(B, C, H, W), e(B, 2, H, W), e(C + 2),
                                                   e(B, H * W), e(B, H * W), None)
        assert gx1.shape == (B, C, H, W) and gx2.shape == (B, 2, H, W) and gw.shape == (C + 2,)
        assert O.pw_gemm_dgrad(e(B, 5, H, W), e(5, C, 1, 1), None, 0, None, 3).shape == (B, C, H, W)
        gw, gb = O.pw_gemm_wgrad(e(B, 5, H, W), e(B, C, H, W), True, None, None, 3)
        assert gw.shape == (5, C) and gb.shape == (5,)
        gb, gmap = O.bias_grads(e(B, 5, H, W), False, True)
        assert gb.numel() == 0 and gmap.shape == (5, H, W)
        y, z, am = O.pointwise(e(B, C, H, W), e(5, C, 1, 1), e(5), None, None, 1, None, 0, False, None, None, True, 3, None)
        assert y.shape == z.shape == (B, 5, H, W) and am.numel() == 0
        y, z, am = O.pointwise(e(B, C, H, W), e(5, C, 1, 1), e(5), None, None, 1, None, 0, False, None, None, False, 2, None)
        assert z.numel() == 0 and am.numel() == 1024     # the f16x2 scheme's amax words are an explicit op output
        y, z, am = O.pointwise(e(B, C, H, W), e(5, C, 1, 1), e(5), None, e(B, 5, H, W), 1, None, 0, False, None, None, True,
                               3, e(5))                          # gated epilogue (blend with the residual)
        assert y.shape == (B, 5, H, W)
        gh, gadv, ga = O.gated_blend_backward_out(e(B, C, H, W), e(B, C, H, W), e(B, C, H, W), e(C))
        assert gh.shape == gadv.shape == (B, C, H, W) and ga.shape == (C,)
        gx, gw, gb = O.dwconv_geo_bwd(e(B, C, H, W), e(B, C, H, W), e(C, 1, 5, 5), e(B, C, H, W), True)
        assert gx.shape == (B, C, H, W) and gw.shape == (C, 1, 5, 5) and gb.shape == (C,)
        assert O.dwconv_geo_dgrad_add(e(B, C, H, W), e(C, 1, 5, 5), e(B, C, H, W)).shape == (B, C, H, W)
        assert O.concat_channels([e(B, 3, H, W), e(B, 4, H, W)]).shape == (B, 7, H, W)
        assert O.slice_channels(e(B, 7, H, W), 3, 4).shape == (B, 4, H, W)
        loss, grad = O.paradis_loss(e(B, C, H, W), e(B, C, H, W), e(C), e(H), 1, 1.0, True)
        assert loss.shape == () and grad.shape == (B, C, H, W)


def test_direct_kernel_calls_only_for_plain_untraced_tensors():
    """The eager front ends call a kernel's Python function directly (``ops.RAW``) only where nothing could be watching
    the dispatcher: a real ``torch.Tensor``, grad mode off (inside ``autograd.Function`` forward / a first-order
    backward), no TorchDispatchMode (FakeTensorMode, ``make_fx`` - also with real tensors), no functorch transform."""
    from torch.fx.experimental.proxy_tensor import make_fx
    t = torch.randn(3)
    assert not ops._plain(t)                        # grad mode on: a recording call, or create_graph=True
    seen = []
    with torch.no_grad():
        assert ops._plain(t)
        assert not ops._plain(torch.nn.Parameter(t))
        make_fx(lambda x: (seen.append(ops._plain(x)), x * 2)[1], tracing_mode="real")(t)
        torch.func.vmap(lambda x: (seen.append(ops._plain(x)), x.sum())[1])(torch.randn(2, 3))
        with FakeTensorMode():
            seen.append(ops._plain(torch.empty(2)))
    assert seen == [False, False, False]
    assert set(ops.RAW) == set(ops.OPS)
