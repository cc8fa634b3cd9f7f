"""Diagnostic: parity and timing of the W = 64 advection kernels for the library named by PARADIS_HIP_LIB
(A/B of build/variants/lib_<name>.so on one box: tools/adv_ab.sh).  Prints one line.
    python tools/adv_ab.py [vel_scale ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import paradis_oracle as O
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids


def rms_rel(a, b):
    d = a.double() - b.double()
    return float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())


def parity(H, W, poles, scale, mode="bicubic"):
    B, K = 2, 4
    _, lg, og = make_grids(H, W, poles)
    g = torch.Generator().manual_seed(5)
    f, ct = torch.randn(B, K, H, W, generator=g), torch.randn(B, K, H, W, generator=g)
    u, v = torch.randn(B, K, H, W, generator=g) * scale, torch.randn(B, K, H, W, generator=g) * scale
    fd, ud, vd = (t.double().requires_grad_(True) for t in (f, u, v))
    yr = O.sl_advect_core(fd, ud, vd, 0.196887, O.GridGeometry(lg.double(), og.double()), mode)
    yr.backward(ct.double())
    f32, u32, v32 = (t.clone().requires_grad_(True) for t in (f, u, v))
    y32 = O.sl_advect_core(f32, u32, v32, 0.196887, O.GridGeometry(lg, og), mode)
    y32.backward(ct)
    fc, uc, vc = (t.cuda().requires_grad_(True) for t in (f, u, v))
    y = ops.sl_advect(fc, uc, vc, ops.AdvectGeometry(lg, og), 0.196887, mode)
    y.backward(ct.cuda())
    return (rms_rel(y.detach().cpu(), yr.detach()), rms_rel(y32.detach(), yr.detach()),
            rms_rel(fc.grad.cpu(), fd.grad), rms_rel(f32.grad, fd.grad),
            rms_rel(uc.grad.cpu(), ud.grad), rms_rel(u32.grad, ud.grad))


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    scales = [float(a) for a in sys.argv[1:]] or [0.05, 1.0]
    name = os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped"))
    pr = ["%.1e/%.1e f %.1e/%.1e u %.1e/%.1e" % parity(32, 64, False, 1.0), "%.1e/%.1e f %.1e/%.1e u %.1e/%.1e" % parity(33, 64, True, 0.3)]
    B, K, H, W = 32, 768, 32, 64
    _, lg, og = make_grids(H, W, False)
    geom = ops.AdvectGeometry(lg, og)
    f = torch.randn(B, K, H, W, device="cuda")
    go = torch.randn(B, K, H, W, device="cuda")
    x = torch.randn(64 << 20, device="cuda")
    for _ in range(2000):
        x = x * 1.0001
    out = []
    for sc in scales:
        vel = torch.randn(B, 2 * K, H, W, device="cuda") * sc
        args = ops._geom_args(geom, f.device, 0.196887 / 8, "bicubic", None)
        with torch.no_grad():
            tf = timed(lambda: ops._sl_advect_vel(f, vel, *args), 200)
            tb = timed(lambda: ops._sl_advect_vel_backward(go, f, vel, *args), 100)
        out.append("scale %g: fwd %.1f us (%.3f) bwd %.1f us (%.3f)" % (sc, tf, 805.3 / tf / 8e3, tb, 1409.3 / tb / 8e3))
    print("%-14s | %s | parity(gpu/cpu32 vs fp64) 32x64: %s ; 33x64 poles: %s" % (name, " ; ".join(out), pr[0], pr[1]), flush=True)


if __name__ == "__main__":
    main()
