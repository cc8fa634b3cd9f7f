#!/usr/bin/env python3
"""Benchmark of the PARADIS ADR hot path on MI355X (contract: see the task description).

Metric (BASELINE.json): training samples/sec (whole node) on the 5.625 deg ERA5 grid.
One "step" = forward (S=1 rollout step) + ParadisLoss + backward + AdamW on one synthetic
ERA5-shaped batch of 32 samples per GPU (configs[1]: 32x64 grid, default 60 M-parameter model,
fp32).  N>1: one process per GPU (torchrun), batch-sharded DDP over RCCL, weak scaling.

Prints ONE JSON line on rank 0.  `value` / `ms_per_step` / `roofline` are measured with the default,
reference-width GEMM arithmetic: fp32 in / fp32 accumulate / fp32 out through the exact three-term bf16
decomposition of every operand value (bf16x3: 24 significand bits, fp32's exponent range per element, six
products on the bf16 MFMA; `--gemm exact` = the f32 MFMA chain).  `roofline` = the pointwise GEMMs (dominant
kernels), timed live with HIP events on the launch stream.  Two more timed legs of the same K steps are
reported beside the headline and never as the headline: `exact_f32_gemm` (v_mfma_f32_32x32x2_f32) and
`f16x2_emulated` (opt-in block-exponent format, NOT reference-width arithmetic).  `cpu_baseline` = the CPU
oracle (port of the reference path), timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak (no xf32 on gfx950)
# bf16-split GEMMs execute 6 bf16 partial products per fp32 product on the 2.5 PFLOP/s dense bf16 pipe:
# the ceiling in algorithmic (fp32-equivalent) FLOP/s is a sixth of it
MFMA_BF16_PEAK_TFLOPS = 2500.0
SPLIT_PRODUCTS = 6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak

WORKLOADS = {
    # name: (nlat, nlon, poles, per-GPU batch, rollout steps)
    "era5_5.625deg_32x64_S1_B32": (32, 64, False, 32, 1),
    "era5_5.625deg_32x64_S6_B32": (32, 64, False, 32, 6),
    "era5_1.4deg_128x256_S1_B8": (128, 256, False, 8, 1),
    "era5_0.25deg_721x1440_fwd_B1": (721, 1440, True, 1, 1),     # forward only (implies --forward-only)
}


def _library_sha256():
    import hashlib
    from paradis_model_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


# suffix of the profile tags of each workload (tools/profile_round.sh <tag> [bench args]): profiles/r03b_* is the
# default workload, profiles/r03a_cfg3_* the 128x256 one ...
# (the S = 6 rollout launches the kernels of the S = 1 step six times over, same shapes: it reads the default profile)
PROFILE_SUFFIX = {"era5_5.625deg_32x64_S1_B32": "", "era5_5.625deg_32x64_S6_B32": "",
                  "era5_1.4deg_128x256_S1_B8": "_cfg3", "era5_0.25deg_721x1440_fwd_B1": "_cfg4"}
WORKLOAD = "era5_5.625deg_32x64_S1_B32"     # set by main()


def _profile(kind: str):
    """Newest committed PMC summary of `kind` FOR THE WORKLOAD BEING RUN (profiles/<tag>[_cfgN]_<kind>.json, written by
    tools/profile_round.sh + tools/summarize_profile.py for this same bench command under rocprofv3 --pmc) with its
    provenance: the summary records the sha256 of the library it profiled; a summary of another build is still
    reported but marked, and one whose kernels no longer exist in the loaded library is dropped.
    Returns (data, source) or (None, None)."""
    import glob
    import re
    pat = re.compile(r"^r\d+[a-z]*%s_%s\.json$" % (re.escape(PROFILE_SUFFIX.get(WORKLOAD, "_none")), re.escape(kind)))
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", f"*_{kind}.json"))
                    if pat.match(os.path.basename(f))), key=lambda f: (os.path.basename(f).split("_")[0], os.path.getmtime(f)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        data = json.load(f)
    meta = data.pop("_meta", {}) if isinstance(data, dict) else {}
    src = {"from_profile": os.path.relpath(files[-1], ROOT), "profiled_commit": meta.get("commit"),
           "same_library_build": meta.get("library_sha256") == _library_sha256() if meta.get("library_sha256") else None}
    return data, src


def _kernels_exist(names) -> bool:
    """every profiled kernel name must still be a symbol of the loaded library"""
    from paradis_model_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        blob = f.read()
    return all(n.split("<")[0].split("(")[0].strip().encode() in blob for n in names)


def pmc_mfma_busy():
    """Matrix-pipe busy fraction of the GEMM kernels (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 *
    1024 SIMDs)) from the committed PMC pass; NOT measured in this run: `source` says where it comes from."""
    data, src = _profile("mfma_busy")
    if not data:
        return None
    vals = {k: v for k, v in data.items() if "gemm" in k}
    if not vals or not _kernels_exist(vals):
        return None
    return {"values": vals, "source": src}


def pmc_traffic(substr: str, per_call: bool = False):
    """HBM bytes per launch of the kernels whose name contains `substr` from the committed rocprofv3 PMC
    passes (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 2x read-side correction of
    MI355X_MICROARCH.md); launch-weighted mean - or, `per_call`, per CALL of an op that launches several of
    them (the windowed advection: strip kernel(s) + the fix-up of the deferred points): the bytes of all matching
    kernels over the launches of the most frequent one.  Counters cannot be collected inside a timed run, so
    this is a read-back: returns (bytes, source) with the provenance, (None, None) without a usable profile."""
    data, src = _profile("traffic")
    if not data:
        return None, None
    num = den = 0.0
    names = []
    for name, rec in data.items():
        if substr in name and "hbm_bytes_per_launch" in rec:
            num += rec["hbm_bytes_per_launch"] * rec.get("launches", 1)
            den = max(den, rec.get("launches", 1)) if per_call else den + rec.get("launches", 1)
            names.append(name)
    if not den or not _kernels_exist(names):
        return None, None
    src = dict(src, kernels=names)
    return num / den, src


def usable_cores() -> int:
    """CPU cores this process may really use: min(affinity mask, cgroup cpu.max quota).  The GPU
    boxes expose 256 logical CPUs but cap the container at 16 via cgroup; using more threads than
    the quota oversubscribes and slows the oracle >10x (measured, tools/cpu_threads.py)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


T_START = time.perf_counter()


def _mem_available_gb() -> float:
    """host memory this process may still take: MemAvailable, capped by the cgroup's memory.max - memory.current"""
    avail = 0.0
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable"):
                    avail = int(ln.split()[1]) / 1e6
    except (OSError, ValueError):
        return 0.0
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            avail = min(avail, (int(lim) - int(open("/sys/fs/cgroup/memory.current").read())) / 1e9)
    except (OSError, ValueError):
        pass
    return avail


def cpu_model() -> str:
    """CPU model string of the box ('core count and CPU model printed')"""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cfg, nlat, nlon, poles, batch, steps_timed=2, warmup=3):
    """Oracle (CPU restatement of the reference path: same ATen/oneDNN operators) training step
    on the host cores: fwd + loss + bwd + AdamW, fp32."""
    torch.set_num_threads(usable_cores())
    from oracle import paradis_oracle as O
    from paradis_model_amd.config import feature_layout, stub_datamodule
    from paradis_model_amd.harness import make_grids, synthetic_batch, assemble_model_input
    from paradis_model_amd.model import Paradis

    lay = feature_layout(cfg)
    lat_deg, lg, og = make_grids(nlat, nlon, poles)
    torch.manual_seed(42)
    holder = Paradis(stub_datamodule(cfg), cfg, lg, og)          # parameters only (CPU), never run
    params = {k: v.detach().clone().requires_grad_(True) for k, v in holder.state_dict().items()}
    del holder
    spec = O.spec_from_cfg(cfg, nlat, nlon, lay.num_in_dyn_features, lay.num_in_static_features,
                           lay.num_out_features)
    fw = O.feature_weights(torch.tensor([1.0] * 83 + [0.1] * 13 + [1.0]),
                           torch.tensor(cfg.features.pressure_levels), 97, 6)
    lw = O.latitude_weights(lat_deg)
    o = cfg.training.optimizer
    opt = torch.optim.AdamW(list(params.values()), lr=o.lr, weight_decay=o.weight_decay,
                            betas=(o.beta1, o.beta2))
    inp, tgt, forc, const = synthetic_batch(nlat, nlon, poles, batch, 1)
    mi = assemble_model_input(inp, forc.permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1),
                              const[:, :1].permute(0, 1, 4, 2, 3))

    def step():
        opt.zero_grad(set_to_none=True)
        y = O.paradis_forward(params, spec, mi, lg, og, interp_impl="aten_ref")
        O.paradis_loss(y, tgt[:, 0], fw, lw).backward()
        opt.step()

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(steps_timed):
        step()
    dt = (time.perf_counter() - t0) / steps_timed
    return {"value": batch / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "kind": "port", "batch": batch, "s_per_step": dt,
            "sample": f"{steps_timed} train steps at batch {batch} on the same "
                      f"{nlat}x{nlon} S=1 workload ({warmup} warm-up), {dt:.2f} s/step"}


def _rooflines(s, elapsed_s, gemm):
    """roofline records of one timed region from the LaunchProfiler summary `s` (GEMMs: algorithmic FLOP against the
    dense bf16 peak / 6 resp. the f32 MFMA peak; advection: 16 / 28 algorithmic bytes per gather point against HBM)"""
    out = {}
    gem = [s[k] for k in ("pw_gemm_fwd", "pw_gemm_dgrad", "pw_gemm_wgrad") if k in s]
    if gem:
        flops = sum(g["work"] for g in gem)
        ms = sum(g["ms"] for g in gem)
        n = sum(g["launches"] for g in gem)
        ach = flops / (ms * 1e-3) / 1e12
        products = {"f16x2": 3, "bf16x3": SPLIT_PRODUCTS, "bf16": 1}.get(gemm)
        if gemm == "bf16":     # the bf16-mixed (autocast) leg: ONE product per multiply, the dense bf16 peak itself
            kname = ("pw_gemm_bf16 kernels (fwd+dgrad+wgrad; operands rounded to bf16, 1 x v_mfma_f32_32x32x16_bf16 per "
                     "product, fp32 accumulate) - NOT reference-width for the fp32 parity path")
            peak = MFMA_BF16_PEAK_TFLOPS
        elif gemm == "bf16x3":
            kname = ("pw_gemm_split_wide_kernel<2, 3>/pw_gemm_wgrad_split_kernel<3> (fwd+dgrad+wgrad; fp32 operands as "
                     "3 bf16 terms, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate)")
            peak = MFMA_BF16_PEAK_TFLOPS / products
        elif gemm == "f16x2":
            kname = ("pw_gemm_split_wide_kernel<2>/pw_gemm_wgrad_split_kernel<2> (fwd+dgrad+wgrad; fp32 operands as "
                     "2 f16 terms, 3 x v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate)")
            peak = MFMA_BF16_PEAK_TFLOPS / products
        else:
            kname = "pw_gemm_dma_kernel/pw_gemm_kernel (fwd+dgrad+wgrad, v_mfma_f32_32x32x2_f32)"
            peak = MFMA_F32_PEAK_TFLOPS
        # (the committed PMC passes profiled the default arithmetic: no read-back for the bf16-mixed leg)
        gemm_traffic, gemm_src = (None, None) if gemm == "bf16" else pmc_traffic("pw_gemm")
        out["roofline"] = {"kernel": kname,
                           "bound": "mfma", "achieved": ach, "peak": peak,
                           "unit": "TFLOP/s", "frac": ach / peak,
                           "flops": "algorithmic 2*M*N*K per GEMM (fp32-equivalent)"
                                    + (f"; executed 16-bit MFMA rate = {products}x achieved, peak = 2500/{products}"
                                       if products and products > 1 else ""),
                           "traffic": gemm_traffic, "traffic_source": gemm_src,
                           "launches": n, "avg_launch_ms": ms / n,
                           "flops_per_launch": flops / n,
                           "share_of_step": ms / (1e3 * elapsed_s),
                           "mfma_busy_pmc": None if gemm == "bf16" else pmc_mfma_busy()}
    for key, name in (("sl_advect_fwd", "roofline_advect_fwd"), ("sl_advect_bwd", "roofline_advect_bwd")):
        if key in s:
            r = s[key]
            ach = r["work"] / (r["ms"] * 1e-3) / 1e9
            tr, tr_src = pmc_traffic(key, per_call=True)
            out[name] = {"kernel": key, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": tr_src,
                         "launches": r["launches"], "avg_launch_ms": r["ms"] / r["launches"],
                         "bytes_per_launch": r["work"] / r["launches"]}
    return out


def other_config_leg(name, n_steps, dev, gemm):
    """One bounded leg of another BASELINE config in the same process (verdict r4 item 2): fresh default model on that
    grid, one warm-up step, `n_steps` timed steps (training step, or the inference forward for the 0.25-degree config)
    with the launch events on, its own peak memory and advection / GEMM rooflines; PMC traffic is read back from the
    committed profile of THAT workload (profiles/*_cfg{2,3,4}_traffic.json)."""
    global WORKLOAD
    import gc
    from paradis_model_amd import _lib
    from paradis_model_amd.config import default_config, feature_layout, stub_datamodule
    from paradis_model_amd.harness import TrainStep, assemble_model_input, make_grids, synthetic_batch
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis

    nlat, nlon, poles, B, S = WORKLOADS[name]
    fwd_only = "_fwd_" in name
    keep = WORKLOAD
    WORKLOAD = name
    try:
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        cfg = default_config()
        lay = feature_layout(cfg)
        lat_deg, lg, og = make_grids(nlat, nlon, poles)
        torch.manual_seed(cfg.init.seed)
        model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
        batch = synthetic_batch(nlat, nlon, poles, B, S, seed=1234, device=dev)
        if fwd_only:
            mi = assemble_model_input(batch[0], batch[2].permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1),
                                      batch[3][:, :1].permute(0, 1, 4, 2, 3))

            def step(_b):
                with torch.no_grad():
                    return model(mi).mean()
        else:
            step = TrainStep(model, build_loss(cfg, lat_deg).to(dev), cfg, num_common=lay.num_common_features,
                             n_inputs=cfg.dataset.n_time_inputs)
        step(batch)                       # warm-up (allocator pools, weight images)
        torch.cuda.synchronize()
        prof = _lib.LaunchProfiler()
        _lib.PROFILER = prof
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loss = step(batch)
        torch.cuda.synchronize()
        e = time.perf_counter() - t0
        _lib.PROFILER = None
        rec = {"value": B * n_steps / e, "unit": "samples/s", "ms_per_step": 1e3 * e / n_steps, "steps": n_steps,
               "warmup": 1, "grid": f"{nlat}x{nlon}", "rollout_steps": S, "per_gpu_batch": B,
               "mode": "forward-only" if fwd_only else "train", "gemm_arithmetic": gemm,
               "peak_hbm_gb": torch.cuda.max_memory_allocated(dev) / 1e9, "final_loss": float(loss)}
        rec.update(_rooflines(prof.summary(), e, gemm))
        return rec
    except Exception as exc:              # never lose the headline line over an extra leg
        _lib.PROFILER = None
        return {"error": repr(exc)[:300]}
    finally:
        WORKLOAD = keep
        model = batch = step = prof = None
        gc.collect()
        torch.cuda.empty_cache()


OTHER_CONFIGS = (("era5_5.625deg_32x64_S6_B32", 2), ("era5_1.4deg_128x256_S1_B8", 3), ("era5_0.25deg_721x1440_fwd_B1", 3))


def _launch_command(n_gpus: int, argv, port: int):
    """torchrun command line for N ranks of this file on ONE node (the form the task contract names:
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)."""
    child_argv = [a for a in argv if a != "--launch-only"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *child_argv]


def _self_launch(args) -> int:
    """Parent of a bare `python bench.py --gpus N`: spawns the ranks with torch.distributed.run as a CHILD process
    (reference train.py:49: Lightning's strategy="ddp" does the same per-GPU process launch), relays the one JSON line
    rank 0 printed and returns the launcher's return code.  Runs before any GPU initialisation: counting devices
    (`torch.cuda.device_count()`) does not create a HIP context."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = _launch_command(args.gpus, sys.argv[1:], port)
    if args.launch_only:
        print(json.dumps({"launch": cmd, "n_gpus": args.gpus, "master_addr": "127.0.0.1", "master_port": port}))
        return 0
    share = os.environ.get("PARADIS_SHARE_GPU0") == "1"      # test hook: all ranks on cuda:0 over gloo
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s)\n")
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    env["PARADIS_BENCH_LAUNCHER"] = "bench.py self-launch (torch.distributed.run child)"
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)   # stderr: inherited, streams through
    line = None
    for ln in proc.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            sys.stderr.write(ln + "\n")                      # launcher chatter never reaches our stdout
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif proc.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        return 1
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="era5_5.625deg_32x64_S1_B32", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2,
                    help="batch of the bounded CPU sample (2 = the reference's own CPU case, configs[0]; "
                         "measured best samples/s on the 16-core box)")
    ap.add_argument("--forward-only", action="store_true",
                    help="inference forward under no_grad (BASELINE configs[4]); value = samples/s")
    ap.add_argument("--checkpoint", action="store_true", help="per-layer activation checkpointing")
    ap.add_argument("--optimizer", default="adamw", choices=["adamw", "muon", "normuon"],
                    help="adamw = the measured configuration (SURVEY 8d); normuon = the reference's shipped default")
    ap.add_argument("--gemm", default=os.environ.get("PARADIS_GEMM", "bf16x3"),
                    choices=["bf16x3", "split", "exact", "f16x2"],
                    help="pointwise GEMM arithmetic of the HEADLINE leg (all fp32 in/accumulate/out): bf16x3 (= split, "
                         "default) = exact three bf16 terms per value, 6 products on the bf16 MFMA; exact = f32 MFMA; "
                         "f16x2 = opt-in block-exponent emulation (two f16 terms of the per-tensor scaled operands) - "
                         "not reference-width arithmetic, labelled as such in the output")
    ap.add_argument("--no-extra-legs", "--no-exact-leg", dest="no_extra_legs", action="store_true",
                    help="skip the extra timed loops (exact_f32_gemm, f16x2_emulated)")
    ap.add_argument("--amp", action="store_true",
                    help="run the HEADLINE leg in the reference's bf16-mixed mode (torch.autocast(bfloat16): one-product bf16 "
                         "GEMMs) - a diagnostic: NOT reference-width arithmetic, labelled as such; skips the extra legs")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bounded legs of the other BASELINE configs (other_configs: S = 6 rollout, 128x256 B = 8, "
                         "721x1440 forward)")
    ap.add_argument("--bucket-mb", type=int, default=32, help="DDP gradient bucket size (N>1)")
    ap.add_argument("--static-graph", action="store_true", help="DDP static_graph=True (N>1)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole training step from a captured HIP graph (harness.GraphedTrainStep; with N > 1 "
                         "the DDP step including its RCCL bucket all-reduces); "
                         "implies --no-kernel-events and skips the extra legs")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the HIP-event timing of the GEMM/advection launches")
    ap.add_argument("--launch-only", action="store_true",
                    help="with --gpus N > 1 and no torchrun environment: print the launcher command line that would start "
                         "the N ranks (one JSON line) and exit; nothing touches a GPU")
    ap.add_argument("--ddp-leg-child", action="store_true", help=argparse.SUPPRESS)    # (the ddp_overhead leg, in a process of its own)
    args = ap.parse_args()
    if args.ddp_leg_child:
        args.no_cpu_baseline = args.no_other_configs = args.no_kernel_events = args.no_extra_legs = True

    # `python bench.py --gpus N` called bare (no torchrun environment): start the N ranks OURSELVES, as fresh child
    # processes, before anything in this process touches the GPU (the parent never initialises HIP, so there is no
    # exec of a GPU-initialised process), relay rank 0's one JSON line, exit with the launcher's return code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args))
    if args.launch_only:
        raise SystemExit("--launch-only: nothing to launch (--gpus 1, or already inside a torchrun environment)")

    # The contract is ONE JSON line on stdout.  Libraries write to the C-level stdout behind Python's back (RCCL
    # announces "Librccl path : ..." when a process group initialises): everything but the final line goes to stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    from paradis_model_amd import _lib, ops
    from paradis_model_amd.config import default_config, feature_layout, stub_datamodule
    from paradis_model_amd.harness import (TrainStep, barrier, init_distributed, make_grids,
                                           max_over_ranks, synthetic_batch, wrap_ddp)
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.gemm == "split":
        args.gemm = "bf16x3"
    ops.GEMM_SCHEME = ops._SCHEMES[args.gemm]
    rank, local, world = init_distributed(capturable=args.graph)   # (the async NCCL error handler stays on for eager DDP)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun")
    if os.environ.get("PARADIS_SHARE_GPU0") == "1":   # test hook: all ranks on cuda:0 (with gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distinct_devices = 1
    if world > 1 and os.environ.get("PARADIS_SHARE_GPU0") != "1":
        # one process per GPU: every rank must own a distinct device of this node
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (os.uname().nodename, torch.cuda.current_device()))
        if len(set(ids)) != world:
            raise SystemExit(f"ranks share a device: {ids}")
        distinct_devices = len(set(ids))

    global WORKLOAD
    WORKLOAD = args.workload
    nlat, nlon, poles, B, S = WORKLOADS[args.workload]
    if "_fwd_" in args.workload:
        args.forward_only = True   # 0.25 deg is an inference configuration (BASELINE configs[4])
    if args.batch:
        B = args.batch
    cfg = default_config()
    cfg.compute.gradient_checkpointing = bool(args.checkpoint)
    cfg.training.optimizer.name = args.optimizer
    lay = feature_layout(cfg)
    lat_deg, lg, og = make_grids(nlat, nlon, poles)
    torch.manual_seed(cfg.init.seed)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    loss_fn = build_loss(cfg, lat_deg).to(dev)
    if args.amp:
        args.no_extra_legs = True
    if args.graph:
        if args.optimizer != "adamw":
            raise SystemExit("--graph: AdamW only")
        args.no_kernel_events = args.no_extra_legs = True
    # --graph with N > 1: the DDP step, bucket all-reduces included, is captured (wrapper built on a side stream,
    # 11 eager warm-up iterations inside GraphedTrainStep)
    ddp = wrap_ddp(model, device_ids=[local], bucket_cap_mb=args.bucket_mb, static_graph=args.static_graph,
                   capturable=args.graph)
    step = TrainStep(ddp, loss_fn, cfg, num_common=lay.num_common_features,
                     n_inputs=cfg.dataset.n_time_inputs, capturable=args.graph, amp=args.amp)
    batch = synthetic_batch(nlat, nlon, poles, B, S, seed=1234 + rank, device=dev)
    if args.graph and not args.forward_only:
        from paradis_model_amd.harness import GraphedTrainStep
        step = GraphedTrainStep(step, batch, warmup=2)
    if args.forward_only:
        from paradis_model_amd.harness import assemble_model_input
        mi = assemble_model_input(batch[0], batch[2].permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1),
                                  batch[3][:, :1].permute(0, 1, 4, 2, 3))

        def step(_b, _m=model, _x=mi):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(args.amp)):
                return _m(_x).mean()

    for _ in range(args.warmup):
        step(batch)
    torch.cuda.synchronize()

    prof = None if args.no_kernel_events else _lib.LaunchProfiler()
    _lib.PROFILER = prof
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(batch)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.PROFILER = None
    elapsed = max_over_ranks(elapsed, dev)
    peak_headline_gb = torch.cuda.max_memory_allocated(dev) / 1e9      # (before the extra legs allocate theirs)
    final_loss = float(loss)

    # The same K steps in the other two arithmetics, timed the same way (>= 5 warm-up steps each) and reported
    # beside the headline under their own names.
    def extra_leg(scheme_name, label):
        ops.GEMM_SCHEME = ops._SCHEMES[scheme_name]
        for _ in range(max(args.warmup, 5)):
            step(batch)
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(batch)
        torch.cuda.synchronize()
        barrier()
        e = max_over_ranks(time.perf_counter() - t1, dev)
        ops.GEMM_SCHEME = ops._SCHEMES[args.gemm]
        return {"value": world * B * args.steps / e, "unit": "samples/s", "ms_per_step": 1e3 * e / args.steps,
                "warmup": max(args.warmup, 5), "gemm_arithmetic": label}

    legs = {}
    if not args.no_extra_legs:
        if args.gemm != "exact":
            legs["exact_f32_gemm"] = extra_leg("exact", "fp32 MFMA (v_mfma_f32_32x32x2_f32)")
        if args.gemm != "f16x2":
            legs["f16x2_emulated"] = extra_leg(
                "f16x2", "NOT reference-width: block-exponent emulation, two f16 terms of the per-tensor scaled "
                         "operands (22 significand bits relative to each tensor's maximum), 3 products on f16 MFMA")

    # The reference's SHIPPED training mode (use_amp: true -> precision="bf16-mixed", config/paradis_settings.yaml:75,
    # train.py:56): forward + loss under torch.autocast(bfloat16), where the pointwise GEMMs run ONE bf16 product
    # (PARADIS_GEMM_BF16).  Never the headline: NOT reference-width arithmetic for the fp32 parity path.  Eager, and
    # replayed from a captured graph (at this step length the eager host path is what bounds slow hosts).
    if not args.no_extra_legs and not args.graph and world == 1 and not args.forward_only and args.optimizer == "adamw":
        try:
            from paradis_model_amd.harness import GraphedTrainStep
            astep = TrainStep(model, loss_fn, cfg, num_common=lay.num_common_features, n_inputs=cfg.dataset.n_time_inputs,
                              amp=True)
            for _ in range(max(args.warmup, 5)):
                astep(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                astep(batch)
            torch.cuda.synchronize()
            e = time.perf_counter() - t1
            rec = {"value": B * args.steps / e, "unit": "samples/s", "ms_per_step": 1e3 * e / args.steps,
                   "warmup": max(args.warmup, 5),
                   "gemm_arithmetic": "NOT reference-width, reference AMP mode: torch.autocast(bfloat16) - pointwise GEMMs "
                                      "with operands rounded to bf16, one product, fp32 accumulate, bf16-rounded results; "
                                      "advection / stencils / norms fp32; tensors fp32 except, inside a block, the ones autocast makes bf16 between "
                                      "a producer and its pointwise consumer (bf16 tensors since round 6; PARADIS_BF16_STORAGE=0: fp32 words)"}
            del astep
            try:
                gstep = GraphedTrainStep(TrainStep(model, loss_fn, cfg, num_common=lay.num_common_features,
                                                   n_inputs=cfg.dataset.n_time_inputs, capturable=True, amp=True),
                                         batch, warmup=2)
                for _ in range(3):
                    gstep(batch)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    gstep(batch)
                torch.cuda.synchronize()
                rec["ms_per_step_graph_replay"] = 1e3 * (time.perf_counter() - t1) / args.steps
                del gstep
            except Exception as exc:
                rec["graph_error"] = repr(exc)[:300]
            legs["bf16_mixed_amp"] = rec
        except Exception as exc:
            legs["bf16_mixed_amp"] = {"error": repr(exc)[:300]}

    # The same step (headline arithmetic) replayed from ONE captured HIP graph: what the host costs disappears
    # (N = 1, AdamW; DDP's bucket hooks are not capturable).  Reported beside the headline, never as the headline.
    if (not args.no_extra_legs and not args.graph and world == 1 and not args.forward_only
            and args.optimizer == "adamw"):
        try:
            from paradis_model_amd.harness import GraphedTrainStep
            gstep = GraphedTrainStep(TrainStep(model, loss_fn, cfg, num_common=lay.num_common_features,
                                               n_inputs=cfg.dataset.n_time_inputs, capturable=True), batch, warmup=2)
            for _ in range(max(args.warmup, 3)):
                gstep(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                gstep(batch)
            torch.cuda.synchronize()
            e = time.perf_counter() - t1
            # host time of ONE step issued into an idle queue (a loop of launches measures the queue's back-pressure,
            # not the host): the captured replay against the eager step
            t2 = time.perf_counter()
            gstep(batch)
            host_graph = time.perf_counter() - t2
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            step(batch)
            host_eager = time.perf_counter() - t2
            torch.cuda.synchronize()
            legs["hip_graph_replay"] = {"value": B * args.steps / e, "unit": "samples/s",
                                        "ms_per_step": 1e3 * e / args.steps,
                                        "host_ms_one_replay": 1e3 * host_graph,
                                        "host_ms_one_eager_step": 1e3 * host_eager,
                                        "gemm_arithmetic": args.gemm,
                                        "what": "forward + ParadisLoss + backward + AdamW captured once "
                                                "(harness.GraphedTrainStep), one graph launch per step"}
            del gstep
        except Exception as exc:   # never lose the headline line over the extra leg
            legs["hip_graph_replay"] = {"error": repr(exc)[:300]}

    # What the data-parallel wrapper costs on ONE GPU: the same step through DistributedDataParallel over a
    # world_size = 1 nccl (RCCL) group - Reducer hooks, bucket copies and the RCCL all-reduce kernels of 240 MB of
    # gradients, no wire - eager and replayed from a captured graph.  (The 1 -> 8 curve needs a node; this leg is the
    # part of it a one-GPU box can measure.)
    # Runs in a CHILD process: a process group brings RCCL's watchdog thread, whose failures (it once aborted the process
    # when it polled an event during a graph capture) cannot be caught by a try block - and the headline line must never be
    # lost over an extra leg.  The child rebuilds the same model and step and prints the leg's record as its one line.
    def ddp_leg():
        import socket
        from paradis_model_amd.harness import GraphedTrainStep
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        d1 = wrap_ddp(model, device_ids=[local], bucket_cap_mb=args.bucket_mb, force=True, capturable=True)
        dstep = TrainStep(d1, loss_fn, cfg, num_common=lay.num_common_features,
                          n_inputs=cfg.dataset.n_time_inputs, capturable=True)

        def timed(fn, n):
            for _ in range(3):
                fn(batch)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn(batch)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / n
        ms_plain = timed(step, args.steps)
        ms_ddp = timed(dstep, args.steps)
        rec = {"ms_per_step_plain": ms_plain, "ms_per_step_ddp_nccl_world1": ms_ddp,
               "ddp_overhead_ms": ms_ddp - ms_plain, "bucket_cap_mb": args.bucket_mb,
               "what": "TrainStep through DistributedDataParallel over a 1-rank nccl (RCCL) group vs the plain step, "
                       "same process (a child of the bench), same box"}
        try:
            gd = GraphedTrainStep(dstep, batch)           # 11 eager DDP iterations, then the capture
            rec["ms_per_step_ddp_graph_replay"] = timed(gd, args.steps)
            t2 = time.perf_counter()
            gd(batch)
            rec["host_ms_one_ddp_replay"] = 1e3 * (time.perf_counter() - t2)
            torch.cuda.synchronize()
            del gd
        except Exception as exc:
            rec["ddp_graph_error"] = repr(exc)[:300]
        del dstep, d1
        torch.distributed.destroy_process_group()
        return rec

    if args.ddp_leg_child:
        try:
            rec = ddp_leg()
        except Exception as exc:
            rec = {"error": repr(exc)[:300]}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        os.close(real_stdout)
        return
    if (not args.no_extra_legs and not args.graph and world == 1 and not args.forward_only
            and args.optimizer == "adamw" and not torch.distributed.is_initialized()):
        try:
            import subprocess
            torch.cuda.empty_cache()
            env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
            cmd = [sys.executable, os.path.abspath(__file__), "--ddp-leg-child", "--steps", str(args.steps), "--warmup", "2",
                   "--gemm", args.gemm, "--bucket-mb", str(args.bucket_mb), "--workload", args.workload]
            if args.batch:
                cmd += ["--batch", str(args.batch)]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                legs["ddp_overhead"] = json.loads(lines[-1])
            else:
                err = r.stderr
                at = max(err.find("terminate called"), err.find("Traceback"), 0)        # (the message, not the frames below it)
                legs["ddp_overhead"] = {"error": f"child rc {r.returncode}: " + (err[at:at + 700] if at else err[-300:])}
        except Exception as exc:
            legs["ddp_overhead"] = {"error": repr(exc)[:300]}

    metric = "training samples/sec (whole node) on 5.625deg ERA5 grid, 1/2/4/8 MI355X"
    try:   # use BASELINE.json's exact wording when the file travels with the repo
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            metric = json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": metric,
        "value": world * B * args.steps / elapsed,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "ddp_efficiency_vs": None,     # never reported here: the driver divides by its own N = 1 run
        "dtype": ("bf16-mixed (NOT reference-width: the reference's AMP mode; one-product bf16 GEMMs; module inputs / outputs, residual stream, parameters fp32)" if args.amp
                  else "f32" if args.gemm != "f16x2" else "f32 storage, f16x2 block-exponent GEMM emulation"),
        "data": "synthetic",
        "config": {"workload": args.workload, "grid": f"{nlat}x{nlon}", "rollout_steps": S,
                   "per_gpu_batch": B, "global_batch": world * B, "parameters": 60038475,
                   "optimizer": args.optimizer, "parallelism": f"dp{world}",
                   "ddp": ({"bucket_cap_mb": args.bucket_mb, "static_graph": bool(args.static_graph),
                            "backend": torch.distributed.get_backend(),
                            "world_size_seen": torch.distributed.get_world_size(),
                            "distinct_devices": distinct_devices,
                            "launcher": os.environ.get("PARADIS_BENCH_LAUNCHER", "external torchrun")}
                           if world > 1 else None),
                   "gemm_arithmetic": args.gemm,
                   "gemm_arithmetic_detail": {
                       "f16x2": "NOT reference-width: two f16 terms of the per-tensor scaled operands (22 significand "
                                "bits relative to the tensor maximum), 3 products on f16 MFMA, fp32 accumulate",
                       "bf16x3": "fp32 operands as the exact 3-term bf16 decomposition (24 significand bits, fp32 exponent "
                                 "range per element), 6 products on bf16 MFMA, fp32 accumulate",
                       "exact": "fp32 MFMA (v_mfma_f32_32x32x2_f32)"}[args.gemm],
                   "mode": "forward-only" if args.forward_only else "train",
                   "hip_graph": bool(args.graph),
                   "activation_checkpointing": bool(args.checkpoint),
                   "peak_hbm_gb": peak_headline_gb,
                   "final_loss": final_loss},
    }
    if prof is not None and rank == 0:
        out.update(_rooflines(prof.summary(), elapsed, "bf16" if args.amp else args.gemm))
    out.update(legs)
    # The other BASELINE configs that fit one GPU, as bounded legs of the same process (each frees memory first):
    # configs[2] (S = 6 rollout), configs[3]'s per-GPU shape (128x256, B = 8) and configs[4] (0.25-degree forward).
    if (rank == 0 and world == 1 and not args.no_extra_legs and not args.no_other_configs and not args.graph
            and args.workload == "era5_5.625deg_32x64_S1_B32" and not args.forward_only and args.optimizer == "adamw"):
        import gc
        del step, ddp, model, batch, loss_fn
        gc.collect()
        torch.cuda.empty_cache()
        out["other_configs"] = {name: other_config_leg(name, n, dev, args.gemm) for name, n in OTHER_CONFIGS}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # configs[0] (B = 2, the reference's own CPU case: >= 3 warm-ups) is the record; the
        # headline's own batch (configs[1], B = 32) as a bounded second leg beside it (1 warm-up + 2 timed steps)
        out["cpu_baseline"] = cpu_baseline(cfg, nlat, nlon, poles, args.cpu_batch, steps_timed=8, warmup=3)
        # the headline's own batch on the CPU: ONE timed step, no warm-up (measured 107 s per step on a 16-core EPYC
        # 9575F box - 6 x the per-sample cost of B = 2: the oracle's B = 32 working set leaves every cache), and only
        # where it cannot stretch the run beyond a few minutes or exhaust host memory
        if not args.no_extra_legs and B != args.cpu_batch and nlat * nlon * B <= 32 * 64 * 32:
            spent = time.perf_counter() - T_START
            if spent > 240 or out["cpu_baseline"]["s_per_step"] > 2.0 or _mem_available_gb() < 160:
                out["cpu_baseline"]["headline_batch_leg"] = {
                    "skipped": f"bounded run: {spent:.0f} s spent so far, {out['cpu_baseline']['s_per_step']:.2f} s per B = "
                               f"{args.cpu_batch} step, {_mem_available_gb():.0f} GB of host memory available"}
            else:
                try:
                    out["cpu_baseline"]["headline_batch_leg"] = cpu_baseline(cfg, nlat, nlon, poles, B, steps_timed=1,
                                                                             warmup=0)
                except Exception as exc:
                    out["cpu_baseline"]["headline_batch_leg"] = {"error": repr(exc)[:300]}
    if world > 1:
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
