"""Distribution, over (input, cotangent) seeds, of the gradient-error ratio GPU / CPU-fp32 on the two-layer full-width
chain at 128x256 (tests/test_hip_configs.py::test_cfg3_default_width_model_128x256_gradients_fp64_protocol), for every
GEMM arithmetic, next to a CPU CONTROL arm: the same CPU-fp32 oracle with the summation order of every pointwise GEMM
permuted (input channels of x and W shuffled consistently - the same real-number function, the same oneDNN kernels,
other roundings).  The control shows what the statistic does under pure reordering within ONE arithmetic.

    python tools/cfg3_grad_dist.py <out.json> [seeds=8] [layers=2] [amp=4]
    (GPU arms: exact, bf16x3 with the shipped library; extra libraries: CFG3_EXTRA_LIBS="tag=path,tag=path")

Per seed and arm: norm-wise error of every parameter gradient against the fp64 oracle divided by the CPU-fp32
oracle's own (median and maximum over the parameters), the worst max-abs ratio, and the number of parameters outside
the bounds of tests/test_hip_model.py::_check_grads_by_fp64_protocol.
"""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tests._util import make_grid, max_rel, rms_rel, seeded  # noqa: E402


def setup(seed, layers, amp):
    from paradis_model_amd.config import default_config
    from tests import test_hip_configs as T
    cfg = default_config()
    cfg.model.num_layers = layers
    H, W = 128, 256
    _, lg, og = make_grid(H, W, False)
    model = T._build(cfg, lg, og, bias_scale=0.05)
    x = T._smooth(seeded(23 + 10 * seed, 1, 186, H, W)) * amp
    x[:, -2], x[:, -1] = lg, og
    ct = T._smooth(seeded(24 + 10 * seed, 1, 97, H, W)) * amp
    return cfg, model, x, ct, lg, og, H, W


def ratios(grads, o):
    """per parameter: (norm-wise ratio, max-abs ratio, fails the test's bounds)"""
    out = {}
    for n, ref in o["g64"].items():
        if n not in grads or float(ref.abs().max()) == 0:
            continue
        gg, gc = grads[n].double(), o["g32"][n].double()
        m_g, m_c, r_g, r_c = max_rel(gg, ref), max_rel(gc, ref), rms_rel(gg, ref), rms_rel(gc, ref)
        out[n] = (r_g / max(r_c, 1e-12), m_g / max(m_c, 1e-12), not (m_g <= 8.0 * m_c + 2e-5 and r_g <= 5.0 * r_c + 5e-5),
                  r_g, r_c)
    return out


def summary(rt):
    rs = sorted(v[0] for v in rt.values())
    ms = sorted(v[1] for v in rt.values())
    worst = max(rt.items(), key=lambda kv: kv[1][0])
    return {"median": rs[len(rs) // 2], "p90": rs[int(0.9 * len(rs))], "max": rs[-1], "max_param": worst[0],
            "maxabs_median": ms[len(ms) // 2], "maxabs_max": ms[-1], "failing": sum(1 for v in rt.values() if v[2])}


def gpu_arm(path, seed, layers, amp, scheme):
    """child process: one GPU arm against the saved oracle gradients"""
    from paradis_model_amd import ops
    ops.GEMM_SCHEME = {"exact": ops.GEMM_EXACT, "bf16x3": ops.GEMM_BF16X3}[scheme]
    o = torch.load(path)
    cfg, model, x, ct, lg, og, H, W = setup(seed, layers, amp)
    got = model(x.cuda())
    (got * ct.cuda()).sum().backward()
    grads = {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}
    s = summary(ratios(grads, o))
    s["fwd"] = max_rel(got.detach().cpu(), o["y32"])
    print("RESULT " + json.dumps(s))


def main():
    if sys.argv[1] == "--gpu-arm":
        gpu_arm(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6])
        return
    out_path = sys.argv[1]
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    layers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    amp = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
    from oracle import paradis_oracle as O
    from tests import test_hip_configs as T
    libs = [("shipped", None)]
    for item in filter(None, os.environ.get("CFG3_EXTRA_LIBS", "").split(",")):
        tag, path = item.split("=")
        libs.append((tag, path))
    table = []
    for seed in range(seeds):
        t0 = time.time()
        cfg, model, x, ct, lg, og, H, W = setup(seed, layers, amp)
        spec = T._spec(cfg, H, W)
        model = model.cpu()
        y32, g32 = T._oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float32)
        y64, g64 = T._oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float64)
        o = {"y32": y32, "g32": g32, "g64": g64}
        # control: permuted summation order of every pointwise GEMM of the CPU-fp32 oracle
        orig = O.pointwise
        perms = {}

        def permuted(xx, weight, bias):
            ci = weight.shape[1]
            if ci not in perms:
                perms[ci] = torch.randperm(ci, generator=torch.Generator().manual_seed(1000 + ci))
            p = perms[ci]
            return orig(xx[:, p].contiguous(), weight[:, p].contiguous(), bias)
        O.pointwise = permuted
        try:
            _, gctl = T._oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float32)
        finally:
            O.pointwise = orig
        cpu_own = sorted(rms_rel(g32[n].double(), g64[n]) for n in g64 if float(g64[n].abs().max()) > 0)
        row = {"seed": seed, "cpu32_own_rms_median": cpu_own[len(cpu_own) // 2], "cpu32_own_rms_max": cpu_own[-1],
               "control": summary(ratios(gctl, o))}
        path = "/tmp/cfg3_dist_oracle.pt"
        torch.save(o, path)
        del model
        for tag, lib in libs:
            for scheme in ("exact", "bf16x3"):
                if lib is not None and scheme == "exact":
                    continue
                env = dict(os.environ)
                if lib:
                    env["PARADIS_HIP_LIB"] = lib
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpu-arm", path, str(seed), str(layers),
                                    str(amp), scheme], env=env, capture_output=True, text=True)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
                if not line:
                    print(r.stdout[-2000:], r.stderr[-2000:])
                    raise SystemExit("GPU arm failed")
                row[f"{scheme}@{tag}"] = json.loads(line[0][7:])
        row["seconds"] = time.time() - t0
        table.append(row)
        print(json.dumps(row), flush=True)
        with open(out_path, "w") as f:
            json.dump({"layers": layers, "amp": amp, "rows": table}, f, indent=1)
    # digest
    arms = [k for k in table[0] if isinstance(table[0][k], dict)]
    print("\narm                      median of medians | medians per seed | max of max | failing per seed")
    for a in arms:
        med = sorted(r[a]["median"] for r in table)
        print("%-24s %6.2f | %s | %6.2f | %s" % (a, med[len(med) // 2], " ".join("%.2f" % r[a]["median"] for r in table),
                                                max(r[a]["max"] for r in table), " ".join(str(r[a]["failing"]) for r in table)))


if __name__ == "__main__":
    main()
