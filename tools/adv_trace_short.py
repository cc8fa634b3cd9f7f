import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
H, W, B, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (721, 1440, 1, 768)
mode = sys.argv[5] if len(sys.argv) > 5 else "bicubic"
_, lg, og = make_grids(H, W, H % 2 == 1)
geom = ops.AdvectGeometry(lg, og)
f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
vel = (torch.randn(B, 2 * K, H, W, device="cuda") * 0.05).requires_grad_(True)
go = torch.randn(B, K, H, W, device="cuda")
for _ in range(4):
    y = ops.sl_advect_vel(f, vel, geom, 0.196887 / 8, mode)
    f.grad = None; vel.grad = None
    y.backward(go)
torch.cuda.synchronize()
