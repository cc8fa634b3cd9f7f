#!/usr/bin/env python3
"""Diagnostic: oracle (CPU port) training-step time vs torch thread count, to pick an honest
cpu_baseline configuration on the GPU box's host."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from paradis_model_amd.config import default_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print("affinity cores:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:  # noqa: BLE001
    print("no cgroup cpu.max", e)
for th in [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64]:
    torch.set_num_threads(th)
    t0 = time.perf_counter()
    r = bench.cpu_baseline(default_config(), 32, 64, False, B, steps_timed=1)
    print(f"threads={th:4d} batch={B} -> {r['value']:.3f} samples/s ({r['sample']}) wall {time.perf_counter() - t0:.1f}s", flush=True)
