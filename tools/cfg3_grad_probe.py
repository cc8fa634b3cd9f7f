"""Diagnostic: the gradient comparison of tests/test_hip_configs.py::test_cfg3_default_width_model_128x256_gradients_fp64_protocol
split in two so that several library builds can be judged against ONE evaluation of the CPU oracles.

  python tools/cfg3_grad_probe.py oracle /tmp/cfg3_oracle.pt [layers [amplitude]]    # CPU fp32 + fp64 oracle gradients (minutes)
  PARADIS_HIP_LIB=... PARADIS_GEMM=... python tools/cfg3_grad_probe.py gpu /tmp/cfg3_oracle.pt [tag]

Prints, per run: forward max-rel vs CPU-fp32, the worst parameter by max-rel against fp64 (GPU, CPU-fp32), the median
and maximum norm-wise ratio GPU / CPU-fp32, and the parameters that fail the test's bound.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from paradis_model_amd.config import default_config  # noqa: E402
from tests._util import make_grid, max_rel, rms_rel, seeded  # noqa: E402
from tests import test_hip_configs as T  # noqa: E402


def setup(layers, amp=4.0):
    cfg = default_config()
    cfg.model.num_layers = layers
    H, W = 128, 256
    _, lg, og = make_grid(H, W, False)
    model = T._build(cfg, lg, og, bias_scale=0.05)
    x = T._smooth(seeded(23, 1, 186, H, W)) * amp
    x[:, -2], x[:, -1] = lg, og
    ct = T._smooth(seeded(24, 1, 97, H, W)) * amp
    return cfg, model, x, ct, lg, og, H, W


def main():
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "oracle":
        layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
        amp = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
        cfg, model, x, ct, lg, og, H, W = setup(layers, amp)
        spec = T._spec(cfg, H, W)
        y32, g32 = T._oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float32)
        y64, g64 = T._oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float64)
        torch.save({"layers": layers, "amp": amp, "y32": y32, "y64": y64, "g32": g32, "g64": g64}, path)
        print("oracle saved", path)
        return
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    o = torch.load(path)
    cfg, model, x, ct, lg, og, H, W = setup(o["layers"], o["amp"])
    got = model(x.cuda())
    (got * ct.cuda()).sum().backward()
    e = max_rel(got.detach().cpu(), o["y32"])
    bad, worst, ratios = [], ("", 0.0, 0.0), []
    for n, p in model.named_parameters():
        ref = o["g64"].get(n)
        if ref is None or float(ref.abs().max()) == 0:
            continue
        gg, gc = p.grad.cpu().double(), o["g32"][n].double()
        m_gpu, m_cpu = max_rel(gg, ref), max_rel(gc, ref)
        r_gpu, r_cpu = rms_rel(gg, ref), rms_rel(gc, ref)
        ratios.append((r_gpu / max(r_cpu, 1e-12), n))
        if not (m_gpu <= 6.0 * m_cpu + 2e-5 and r_gpu <= 4.0 * r_cpu + 1e-5):
            bad.append((n, "%.2e" % m_gpu, "%.2e" % m_cpu, "%.2e" % r_gpu, "%.2e" % r_cpu))
        if m_gpu > worst[1]:
            worst = (n, m_gpu, m_cpu)
    ratios.sort()
    cpu = sorted(max_rel(o["g32"][n].double(), o["g64"][n]) for n in o["g64"] if float(o["g64"][n].abs().max()) > 0)
    print("[%s] L=%d amp=%g cpu32 max-rel vs fp64: median %.2e max %.2e" % (tag, o["layers"], o["amp"], cpu[len(cpu) // 2], cpu[-1]))
    print("[%s] lib=%s gemm=%s fwd %.2e worst %s %.3e (cpu32 %.3e); rms ratio median %.2f max %.2f (%s); failing %d"
          % (tag, os.environ.get("PARADIS_HIP_LIB", "shipped"), os.environ.get("PARADIS_GEMM", "default"), e,
             worst[0], worst[1], worst[2], ratios[len(ratios) // 2][0], ratios[-1][0], ratios[-1][1], len(bad)))
    for b in bad[:6]:
        print("    ", b)


if __name__ == "__main__":
    main()
