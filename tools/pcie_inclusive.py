"""The headline step with the batch handed over from PINNED HOST memory every step (what a data loader does) instead of
resident in HBM: copy on a side stream overlapped with the previous step, and the naive in-stream copy.
  python tools/pcie_inclusive.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd.config import default_config, feature_layout, stub_datamodule   # noqa: E402
from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch            # noqa: E402
from paradis_model_amd.loss import build_loss                                            # noqa: E402
from paradis_model_amd.model import Paradis                                              # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    cfg = default_config()
    lay = feature_layout(cfg)
    lat_deg, lg, og = make_grids(32, 64, False)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    step = TrainStep(model, build_loss(cfg, lat_deg).to(dev), cfg, num_common=lay.num_common_features,
                     n_inputs=cfg.dataset.n_time_inputs)
    host = tuple(t.pin_memory() for t in synthetic_batch(32, 64, False, 32, 1, seed=1))
    nbytes = sum(t.numel() * t.element_size() for t in host)
    resident = tuple(t.to(dev) for t in host)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    ms_res = timed(lambda: step(resident))
    ms_naive = timed(lambda: step(tuple(t.to(dev, non_blocking=True) for t in host)))
    side = torch.cuda.Stream()
    nxt = [None]

    def prefetching():
        cur = nxt[0]
        if cur is None:
            cur = tuple(t.to(dev, non_blocking=True) for t in host)
        with torch.cuda.stream(side):                     # next batch crosses PCIe under this step
            nxt[0] = tuple(t.to(dev, non_blocking=True) for t in host)
        step(cur)
        torch.cuda.current_stream().wait_stream(side)
    ms_pre = timed(prefetching)
    print("batch %.1f MB per step; resident %.2f ms (%.1f samples/s); in-stream copy %.2f ms (%.1f); "
          "prefetched on a side stream %.2f ms (%.1f)" % (nbytes / 1e6, ms_res, 32e3 / ms_res, ms_naive, 32e3 / ms_naive,
                                                            ms_pre, 32e3 / ms_pre))


if __name__ == "__main__":
    main()
