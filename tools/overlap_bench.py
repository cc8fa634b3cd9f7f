#!/usr/bin/env python3
"""Diagnostic: do a compute-bound GEMM and memory-bound kernels overlap when issued on two HIP
streams?  (sequential sum vs concurrent wall time)"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd._lib import dptr, lib

B, P, Co, Ci = 32, 2048, 1024, 1024
dy = torch.randn(B, Co, P, device="cuda"); x = torch.randn(B, Ci, P, device="cuda")
dw = torch.empty(Co, Ci, device="cuda"); ws = torch.empty(80 << 20, device="cuda")
a = torch.randn(B, 1152, P, device="cuda"); ga = torch.randn_like(a); gx = torch.empty_like(a)
w = torch.randn(1152, device="cuda"); mean = torch.randn(B, P, device="cuda"); rstd = torch.rand(B, P, device="cuda") + 0.5
gw = torch.empty(1152, device="cuda"); gb = torch.empty(1152, device="cuda"); ws2 = torch.empty(64 << 20, device="cuda")
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)

def gemm(s, n=4):
    for _ in range(n):
        lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(x), dptr(dw), None, B, Co, Ci, P, Co * P, Ci * P, 0, None, None, dptr(ws), sp(s))
def mem(s, n=8):
    for _ in range(n):
        lib.paradis_channel_norm_bwd(dptr(ga), dptr(a), None, dptr(w), dptr(mean), dptr(rstd), dptr(gx), None, dptr(gw),
                                     dptr(gb), B, 1152, 0, P, 1152 * P, 0, 1152 * P, 0, None, 0, dptr(ws2), sp(s))
        lib.paradis_act_bwd(dptr(ga), dptr(a), dptr(gx), a.numel(), 1, sp(s))
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2):
    gemm(s_main); mem(s_main)
tg = wall(lambda: gemm(s_main)); tm = wall(lambda: mem(s_main))
tb = wall(lambda: (gemm(s_main), mem(s_side)))
tseq = wall(lambda: (gemm(s_main), mem(s_main)))
print(f"gemm {tg:.2f} ms, mem {tm:.2f} ms, sequential {tseq:.2f} ms, two streams {tb:.2f} ms")
