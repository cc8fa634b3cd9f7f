#!/usr/bin/env python3
"""A/B on ONE box, un-profiled: the 32x64 B = 32 training step with the weight-gradient GEMMs in series on the
launching stream (arm A) against the same step with them on the side stream (arm B, ops.WgradSide), eager and
replayed from a captured HIP graph, arms interleaved.  Also checks that both arms produce the same parameters
bit for bit after the same number of steps from the same start (same kernels, same order of every sum)."""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from paradis_model_amd import ops  # noqa: E402
from paradis_model_amd.config import default_config, feature_layout, stub_datamodule  # noqa: E402
from paradis_model_amd.harness import GraphedTrainStep, TrainStep, make_grids, synthetic_batch  # noqa: E402
from paradis_model_amd.loss import build_loss  # noqa: E402
from paradis_model_amd.model import Paradis  # noqa: E402


def build(dev, nlat, nlon, capturable, amp=False):
    cfg = default_config()
    lay = feature_layout(cfg)
    lat_deg, lg, og = make_grids(nlat, nlon, False)
    torch.manual_seed(cfg.init.seed)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    step = TrainStep(model, build_loss(cfg, lat_deg).to(dev), cfg, num_common=lay.num_common_features,
                     n_inputs=cfg.dataset.n_time_inputs, capturable=capturable, amp=amp)
    return model, step


def timed(fn, batch, n, warm=3):
    for _ in range(warm):
        fn(batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn(batch)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--grid", default="32x64")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lag", type=int, default=None)
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--priority", type=int, default=None, help="torch stream priority of the side stream")
    ap.add_argument("--main-priority", type=int, default=None,
                    help="run the whole step on a stream of this priority instead of the default stream")
    ap.add_argument("--arm", choices=["serial", "side"], default=None,
                    help="trace mode: run only this arm, eager, --steps steps after 2 warm-ups (for rocprofv3 --kernel-trace)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.lag is not None:
        ops.WgradSide.LAG = a.lag
    if a.priority is not None:
        ops.WgradSide.PRIORITY = a.priority
    print("stream priority range", torch.cuda.Stream.priority_range(), flush=True)
    if a.main_priority is not None:
        torch.cuda.set_stream(torch.cuda.Stream(priority=a.main_priority))
    nlat, nlon = (int(v) for v in a.grid.split("x"))
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(nlat, nlon, False, a.batch, 1, seed=1234, device=dev)
    res = {"grid": a.grid, "batch": a.batch, "steps": a.steps, "lag": ops.WgradSide.LAG, "amp": a.amp, "eager": [], "graph": [],
           "priority": ops.WgradSide.PRIORITY, "main_priority": a.main_priority}
    if a.arm:
        ops.WgradSide.enabled = a.arm == "side"
        model, step = build(dev, nlat, nlon, False, a.amp)
        print(a.arm, timed(step, batch, a.steps, warm=2), "ms per step (under the tracer)")
        return

    # bitwise equality of the two arms (4 steps each from the same initial state)
    finals = []
    for side in (False, True):
        ops.WgradSide.enabled = side
        model, step = build(dev, nlat, nlon, False, a.amp)
        for _ in range(4):
            loss = step(batch)
        torch.cuda.synchronize()
        finals.append((torch.cat([p.detach().flatten() for p in model.parameters()]).clone(), float(loss)))
        del model, step
    res["bitwise_equal_after_4_steps"] = bool(torch.equal(finals[0][0], finals[1][0]))
    res["losses"] = [finals[0][1], finals[1][1]]
    res["max_abs_diff"] = float((finals[0][0] - finals[1][0]).abs().max())
    print(json.dumps({k: res[k] for k in ("bitwise_equal_after_4_steps", "losses", "max_abs_diff")}), flush=True)
    del finals

    model, step = build(dev, nlat, nlon, False, a.amp)
    for r in range(a.rounds):
        row = {}
        for side in ((False, True) if r % 2 == 0 else (True, False)):
            ops.WgradSide.enabled = side
            row["side" if side else "serial"] = timed(step, batch, a.steps)
        res["eager"].append(row)
        print("eager", json.dumps(row), flush=True)
    del model, step
    if not a.no_graph:
        gs = {}
        for side in (False, True):
            ops.WgradSide.enabled = side
            model, step = build(dev, nlat, nlon, True, a.amp)
            gs[side] = (model, GraphedTrainStep(step, batch, warmup=2))
        for r in range(a.rounds):
            row = {}
            for side in ((False, True) if r % 2 == 0 else (True, False)):
                row["side" if side else "serial"] = timed(gs[side][1], batch, a.steps)
            res["graph"].append(row)
            print("graph", json.dumps(row), flush=True)
    res["peak_hbm_gb"] = torch.cuda.max_memory_allocated(dev) / 1e9
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
