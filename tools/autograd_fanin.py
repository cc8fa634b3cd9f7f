"""Diagnostic: which tensors of the training step receive MORE THAN ONE gradient contribution - every such fan-in is
one `at::add` pass of the autograd engine over the tensor (profiles: `vectorized_elementwise_kernel<CUDAFunctor_add>`)
unless one of the producers takes the other as an addend.  Walks the autograd graph of one forward + loss of the
default 32x64 configuration and prints (consumer node, input index, number of producers, shape).
  python tools/autograd_fanin.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd.config import default_config, stub_datamodule      # noqa: E402
from paradis_model_amd.harness import assemble_model_input, make_grids, synthetic_batch   # noqa: E402
from paradis_model_amd.loss import build_loss                               # noqa: E402
from paradis_model_amd.model import Paradis                                 # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = default_config()
    lat_deg, lg, og = make_grids(32, 64, False)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    loss_fn = build_loss(cfg, lat_deg).to(dev)
    batch = synthetic_batch(32, 64, False, 4, 1, seed=1, device=dev)
    input_data, true_data, forcings, constant_data = batch
    mi = assemble_model_input(input_data, forcings.permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1),
                              constant_data[:, :1].permute(0, 1, 4, 2, 3))
    loss = loss_fn(model(mi), true_data[:, 0])
    fanin = collections.Counter()
    meta = {}
    seen, stack = set(), [loss.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        for nxt, idx in fn.next_functions:
            if nxt is None:
                continue
            fanin[(nxt, idx)] += 1
            meta.setdefault((nxt, idx), []).append(type(fn).__name__)
            stack.append(nxt)
    rows = collections.Counter()
    for (fn, idx), n in fanin.items():
        if n >= 2 and type(fn).__name__ != "AccumulateGrad":
            shape = None
            try:
                shape = tuple(fn._input_metadata[idx].shape) if hasattr(fn, "_input_metadata") else None
            except Exception:
                pass
            rows[(type(fn).__name__, idx, n, shape, tuple(sorted(meta[(fn, idx)])))] += 1
    for (name, idx, n, shape, prods), count in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"{count:3d} x  output {idx} of {name}  <- {n} gradients from {prods}  shape {shape}")


if __name__ == "__main__":
    main()
