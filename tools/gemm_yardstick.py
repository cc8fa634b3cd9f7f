#!/usr/bin/env python3
"""Same-box yardstick for the pointwise GEMMs (verdict r4 item 1a).

A TOOL, never the product: it times the vendor's plain bf16 GEMM (``torch.matmul`` / ``F.linear`` on bf16 operands =
hipBLASLt / rocBLAS) on the eleven (Co, Ci) layer shapes of the default model at N = 65,536 (32x64, B = 32) and
N = 262,144 (128x256, B = 8) columns, INTERLEAVED launch group by launch group with this library's six-product fp32
GEMM (``pw_gemm_split_wide_kernel<2,3>`` through the C ABI) on the same box, and prints what each sustains:

* vendor: PF of bf16 MFMA work (2 Co Ci N / t), best of three operand layouts (W[Co,Ci] @ X[Ci,N]; the same batched
  over samples; F.linear on X^T[N,Ci], the layout the vendor library is tuned for);
* ours: fp32-equivalent TF (2 Co Ci N / t) and the bf16 MFMA work it executes for that (six products: x 6).

The vendor GEMM does ONE product per (m, n, k) on operands that are already bf16 in memory; ours reads fp32 activations,
splits them in registers and does six.  What the comparison says: whether the matrix pipe, fed by a tuned kernel without
the split, sustains more executed bf16 work on these shapes than the shipped kernel does - i.e. how much a better k-loop
could still buy.

    python tools/gemm_yardstick.py [--rounds R] [--reps K] [--json out.json] [--short]
    tools/gemm_yardstick_pmc.sh <tag>     (the same under rocprofv3 --pmc: matrix-pipe busy and effective clock per kernel)
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import dptr, lib, stream_ptr  # noqa: E402

SHAPES = [(1024, 186), (384, 1024), (1536, 384), (768, 1024), (1024, 768), (1024, 1024), (896, 1152),
          (896, 896), (1024, 896), (768, 768), (97, 768)]
GRIDS = [("32x64 B=32", 32, 2048), ("128x256 B=8", 8, 32768)]
BF16X3 = 3


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # microseconds per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5, help="interleaved rounds per shape (median reported)")
    ap.add_argument("--reps", type=int, default=8, help="back-to-back launches per timing")
    ap.add_argument("--json", default=None)
    ap.add_argument("--short", action="store_true", help="three shapes only (profiler passes)")
    a = ap.parse_args()
    st = stream_ptr()
    shapes = [(1024, 1024), (896, 1152), (384, 1024)] if a.short else SHAPES
    # warm the clocks on random data
    w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    for _ in range(50):
        w @ w
    torch.cuda.synchronize()
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "rounds": a.rounds, "reps": a.reps,
           "rows": []}
    for gname, B, P in GRIDS:
        N = B * P
        tot = {"vendor_us": 0.0, "ours_us": 0.0, "flop": 0.0}
        for Co, Ci in shapes:
            wf = torch.randn(Co, Ci, device="cuda") * Ci ** -0.5
            xf = torch.randn(B, Ci, P, device="cuda")
            y = torch.empty(B, Co, P, device="cuda")
            wsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Co, Ci, BF16X3), dtype=torch.uint8, device="cuda")
            lib.paradis_pw_gemm_split_weights(dptr(wf), Co, Ci, 0, BF16X3, dptr(wsp), st)
            wb = wf.to(torch.bfloat16)
            xb = xf.to(torch.bfloat16)                                  # [B, Ci, P]
            x2 = xb.permute(1, 0, 2).reshape(Ci, N).contiguous()        # [Ci, N]
            xt = x2.t().contiguous()                                    # [N, Ci]
            variants = {
                "W@X[Ci,N]": lambda: torch.matmul(wb, x2),
                "W@X[B,Ci,P]": lambda: torch.matmul(wb, xb),
                "linear(X^T[N,Ci],W)": lambda: F.linear(xt, wb),
            }

            def ours():
                lib.paradis_pw_gemm_fwd(dptr(wf), None, dptr(wsp), BF16X3, None, dptr(xf), None, None, None, None, 0,
                                        None, dptr(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st)

            for fn in list(variants.values()) + [ours]:      # library heuristics, code objects, first-touch
                for _ in range(3):
                    fn()
            torch.cuda.synchronize()
            ref = (wf.double() @ xf[0].double())
            err = float((y[0].double() - ref).abs().max() / ref.abs().max())
            ts = {k: [] for k in list(variants) + ["ours"]}
            for r in range(a.rounds):
                order = list(variants.items()) + [("ours", ours)]
                order = order[r % len(order):] + order[:r % len(order)]     # rotate: a fixed order favours the later arm
                for k, fn in order:
                    ts[k].append(timed(fn, a.reps))
            med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
            vk = min(variants, key=lambda k: med[k])
            flop = 2.0 * Co * Ci * N
            row = {"grid": gname, "Co": Co, "Ci": Ci, "N": N, "vendor_layout": vk, "vendor_us": med[vk],
                   "vendor_bf16_PF": flop / med[vk] / 1e9, "ours_us": med["ours"],
                   "ours_f32eq_TF": flop / med["ours"] / 1e6, "ours_exec_bf16_PF": 6 * flop / med["ours"] / 1e9,
                   "ours_err_vs_fp64": err, "all_vendor_us": {k: med[k] for k in variants}}
            out["rows"].append(row)
            tot["vendor_us"] += med[vk]; tot["ours_us"] += med["ours"]; tot["flop"] += flop
            print(f"{gname:12s} Co={Co:5d} Ci={Ci:5d}  vendor bf16 {med[vk]:8.1f} us = {row['vendor_bf16_PF']:5.3f} PF "
                  f"({vk})   ours {med['ours']:8.1f} us = {row['ours_f32eq_TF']:6.1f} TF fp32-eq = "
                  f"{row['ours_exec_bf16_PF']:5.3f} PF executed   ratio exec/vendor "
                  f"{row['ours_exec_bf16_PF'] / row['vendor_bf16_PF']:5.2f}   err {err:.1e}", flush=True)
            del wf, xf, y, wb, xb, x2, xt
            torch.cuda.empty_cache()
        s = {"grid": gname, "vendor_bf16_PF": tot["flop"] / tot["vendor_us"] / 1e9,
             "ours_f32eq_TF": tot["flop"] / tot["ours_us"] / 1e6, "ours_exec_bf16_PF": 6 * tot["flop"] / tot["ours_us"] / 1e9}
        out.setdefault("sums", []).append(s)
        print(f"== {gname}: vendor {s['vendor_bf16_PF']:.3f} PF; ours {s['ours_f32eq_TF']:.1f} TF fp32-eq = "
              f"{s['ours_exec_bf16_PF']:.3f} PF executed ({s['ours_exec_bf16_PF'] / s['vendor_bf16_PF']:.2f} of the vendor's "
              f"rate with six products and the split on top)", flush=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
