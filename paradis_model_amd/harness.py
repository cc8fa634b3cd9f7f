"""Host-side callers of the hot path (e): synthetic ERA5-shaped
batches, the autoregressive training step of the reference's ``LitParadis.training_step``
(reference ``trainer.py:498-587`` + ``_autoregression_next_input`` ``:710-729``) restated without
Lightning, and the data-parallel wrapper (one process per GPU, RCCL all-reduce overlapped with
backward through ``torch.nn.parallel.DistributedDataParallel``).

Everything here is device agnostic plumbing (tensor concatenation, optimiser, collectives); the
compute is inside the model that is passed in.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# ---------------------------------------------------------------------------------- grids / data
GRID_PRESETS = {
    "5.625deg": (32, 64, False),    # WeatherBench2 64x32 equiangular, no pole rows
    "1.40625deg": (128, 256, False),
    "0.25deg": (721, 1440, True),   # with pole rows
}


def make_grids(nlat: int, nlon: int, poles: bool):
    """(lat_deg[H], lat_grid[H,W], lon_grid[H,W]) in fp32 radians, latitude ascending
    (reference data/era5_dataset.py:77-81,178-182)."""
    if poles:
        lat = torch.linspace(-90.0, 90.0, nlat, dtype=torch.float64)
    else:
        d = 180.0 / nlat
        lat = -90.0 + d / 2 + d * torch.arange(nlat, dtype=torch.float64)
    lon = torch.arange(nlon, dtype=torch.float64) * (360.0 / nlon)
    lat32, lon32 = lat.to(torch.float32), lon.to(torch.float32)
    lg, og = torch.meshgrid(torch.deg2rad(lat32), torch.deg2rad(lon32), indexing="ij")
    return lat32, lg.contiguous(), og.contiguous()


def synthetic_batch(nlat: int, nlon: int, poles: bool, batch: int, steps: int, *, seed: int = 1234,
                    device="cpu", n_dyn: int = 166, n_out: int = 97, n_forc: int = 10):
    """ERA5-shaped batch tuple exactly as the reference dataset yields it after collation
    (reference data/era5_dataset.py:379-382): input [B,1,166,H,W] ~ N(0,1), target [B,S,97,H,W] ~
    N(0,1), forcings [B,S,H,W,10] ~ U(-1,1), constants [B,1,H,W,10] = 4 random fields, z-scored
    inverse lon spacing, cos(lat), cos(lon), sin(lon), lat, lon."""
    g = torch.Generator().manual_seed(seed)
    _, lg, og = make_grids(nlat, nlon, poles)
    inp = torch.randn(batch, 1, n_dyn, nlat, nlon, generator=g)
    tgt = torch.randn(batch, steps, n_out, nlat, nlon, generator=g)
    forc = torch.rand(batch, steps, nlat, nlon, n_forc, generator=g) * 2 - 1
    rnd = torch.randn(4, nlat, nlon, generator=g)
    dlon = math.radians(360.0 / nlon)
    inv = 1.0 / (2 * torch.arcsin(torch.cos(lg) ** 2 * math.sin(dlon / 2)).clamp_min(1e-12) * 6371)
    inv = (inv - inv.mean()) / inv.std()
    const = torch.stack([*rnd, inv, torch.cos(lg), torch.cos(og), torch.sin(og), lg, og])
    const = const.permute(1, 2, 0).unsqueeze(0).unsqueeze(0).expand(batch, 1, -1, -1, -1).contiguous()
    return tuple(t.to(device) for t in (inp, tgt, forc, const))


# ---------------------------------------------------------------------------------- rollout
def assemble_model_input(input_data, forcings_step, constants):
    """cat([input[B,1,166,H,W], forcings[B,1,10,H,W], constants[B,1,10,H,W]], 2).squeeze(1)
    (reference trainer.py:534-538)."""
    if input_data.is_cuda:
        from . import ops   # strided block copies by the HIP library (no ATen cat / permute copies)
        return ops.concat_channels([input_data.squeeze(1), forcings_step.squeeze(1), constants.squeeze(1)])
    return torch.cat([input_data, forcings_step, constants], dim=2).squeeze(1)


def next_input(model_input, output, num_common: int, n_inputs: int):
    """Autoregressive channel stack update (reference trainer.py:710-729)."""
    common = output[:, :num_common]
    if n_inputs == 1:
        return common
    if model_input.is_cuda:
        from . import ops
        return ops.concat_channels([model_input[:, num_common:num_common * n_inputs], common])
    return torch.cat([model_input[:, num_common:num_common * n_inputs], common], dim=1)


def rollout_loss(model, loss_fn, batch, *, num_common: int, n_inputs: int, accum: int = 1,
                 detach_every: Optional[int] = None, keep_outputs: bool = False,
                 backward: bool = True, amp: bool = False):
    """The autoregressive hot loop of ``training_step`` (reference trainer.py:508-576): per step
    assemble the input, run the model, accumulate ``loss/(S*accum)``, feed the prediction back;
    ``backward`` at chunk ends (every ``detach_every`` steps and at the last step), detaching the
    carried input.  Returns (sum of chunk losses as a detached tensor, outputs if requested).

    ``amp``: the forward and the loss of every step run under ``torch.autocast(bfloat16)``; the ``backward`` calls run
    OUTSIDE the autocast region, as under Lightning's ``precision="bf16-mixed"`` plugin (reference train.py:56) and
    as the PyTorch AMP recipe prescribes."""
    input_data, true_data, forcings, constant_data = batch
    constants = constant_data[:, :1].permute(0, 1, 4, 2, 3)
    forcings = forcings.permute(0, 1, 4, 2, 3)
    S = true_data.size(1)
    total = torch.zeros((), device=input_data.device)
    chunk = 0.0
    outs = []
    for step in range(S):
        with torch.autocast("cuda" if input_data.is_cuda else "cpu", dtype=torch.bfloat16, enabled=bool(amp)):
            mi = assemble_model_input(input_data, forcings[:, step].unsqueeze(1), constants)
            out = model(mi)
            if keep_outputs:
                outs.append(out.detach())
            chunk = chunk + loss_fn(out, true_data[:, step]) / (S * accum)
            input_data = next_input(mi, out, num_common, n_inputs).unsqueeze(1)
        if (detach_every is not None and (step + 1) % detach_every == 0) or step == S - 1:
            if backward:
                chunk.backward()
            total = total + chunk.detach()
            input_data = input_data.detach()
            chunk = 0.0
    return total, outs


class TrainStep:
    """forward(S rollout steps) + ParadisLoss + backward + AdamW, the unit the benchmark times
    (SURVEY.md section 8d).  AdamW hyper-parameters follow reference trainer.py:327-335."""

    def __init__(self, model, loss_fn, cfg, *, num_common: int = 83, n_inputs: int = 2, fused=None,
                 capturable: bool = False, amp: bool = False):
        self.model, self.loss_fn = model, loss_fn
        # amp: forward and loss under torch.autocast(bfloat16) - the reference's shipped ``use_amp: true`` /
        # ``precision="bf16-mixed"`` (config/paradis_settings.yaml:75, train.py:56); parameters, gradients and the
        # optimiser stay fp32, as with Lightning's mixed-precision plugin
        self.amp = bool(amp)
        self.num_common, self.n_inputs = num_common, n_inputs
        o = cfg.training.optimizer
        params = [p for p in model.parameters() if p.requires_grad]
        on_hip = bool(params and params[0].is_cuda)
        name = o.get("name", "adamw")
        if "muon" in name:
            # reference trainer.py:337-364 (dion.Muon / dion.NorMuon over build_param_groups)
            if not on_hip:
                raise RuntimeError("Muon/NorMuon run on the HIP device only")
            from . import optim
            inner = model.module if hasattr(model, "module") else model
            groups = optim.build_param_groups(inner, lr=o.lr, weight_decay=o.weight_decay, optimizer_name=name)
            cls = {"muon": optim.Muon, "normuon": optim.NorMuon}.get(name)
            if cls is None:
                raise ValueError(f"Optimizer {name} not supported. Choose between normuon|muon")
            self.opt = cls(groups, lr=o.lr, weight_decay=o.weight_decay, betas=(o.beta1, o.beta2),
                           use_triton=True)
        elif on_hip and fused is None:
            from .optim import AdamW   # HIP kernel, torch.optim.AdamW semantics
            self.opt = AdamW(params, lr=o.lr, weight_decay=o.weight_decay, betas=(o.beta1, o.beta2),
                             capturable=capturable)
        else:
            self.opt = torch.optim.AdamW(params, lr=o.lr, weight_decay=o.weight_decay,
                                         betas=(o.beta1, o.beta2), fused=bool(fused))
        self.detach_every = o.get("detach_gradient_every", None)

    def __call__(self, batch):
        self.opt.zero_grad(set_to_none=True)
        loss, _ = rollout_loss(self.model, self.loss_fn, batch, num_common=self.num_common,
                               n_inputs=self.n_inputs, detach_every=self.detach_every, amp=self.amp)
        self.opt.step()
        return loss


class GraphedTrainStep:
    """The whole training step - forward (S rollout steps), ParadisLoss, backward, AdamW: ~2,500 kernel launches -
    captured once into a HIP graph (``torch.cuda.CUDAGraph``) and replayed: the host enqueues one graph launch per
    step instead of walking dispatcher -> autograd -> Python kernel -> ctypes for every op (23.5 ms of host time per
    step at 32x64, the bound below ~4 samples per GPU).  Everything on the path is capture-safe: the ops launch on
    the current stream through the C ABI without host synchronisation, workspaces come from the caching allocator
    (the graph's private pool under capture), and the optimiser's step count and learning rate live on the device
    (``optim.AdamW(capturable=True)``).

    ``step`` must be a ``TrainStep(..., capturable=True)``; ``example_batch`` fixes the shapes.  ``warmup`` eager
    steps run first on a side stream (they are real optimiser steps: kernel attributes, optimiser state and the
    allocator's pools must exist before capture).

    Data parallel: a model wrapped by ``wrap_ddp(..., capturable=True)`` is captured WITH its gradient all-reduces -
    RCCL collectives are capturable, the Reducer launches them from its autograd hooks onto the process group's
    stream, which forks from and joins the capturing stream inside the capture.  What that needs (PyTorch's CUDA-graph
    notes for DDP): the wrapper constructed on a side stream, at least 11 eager DDP iterations before the capture (the
    Reducer rebuilds its buckets after the first one and settles its bookkeeping over the next ones), and no
    asynchronous NCCL error handling thread touching the captured work (TORCH_NCCL_ASYNC_ERROR_HANDLING=0, set by
    ``init_distributed(capturable=True)``)."""

    DDP_WARMUP = 11

    def __init__(self, step: TrainStep, example_batch, warmup: int = 2):
        if not getattr(step.opt, "capturable", False):
            raise ValueError("GraphedTrainStep needs TrainStep(..., capturable=True)")
        if isinstance(step.model, torch.nn.parallel.DistributedDataParallel):
            warmup = max(warmup, self.DDP_WARMUP)
        self.step = step
        self.static_batch = tuple(t.clone() for t in example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step(self.static_batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_steps = max(1, warmup)
        self.graph = torch.cuda.CUDAGraph()
        step.opt.zero_grad(set_to_none=True)
        # With a process group alive, RCCL's watchdog thread polls its work events (hipEventQuery) whenever it likes; under the
        # default capture mode ("global") such a call from ANY thread while this one captures is an error that kills the
        # process ("operation not permitted when stream is capturing": seen once in ~10 runs of the 1-rank nccl test).
        # "thread_local" checks the capturing thread only.
        import torch.distributed as dist
        with_pg = dist.is_available() and dist.is_initialized()
        mode = "thread_local" if with_pg else "global"
        if with_pg:
            # ... and give the watchdog two of its 100 ms rounds to retire the work items of the eager warm-up iterations
            # (all complete: the device is synchronised), so that it holds no event to poll while the capture runs
            import time as _time
            _time.sleep(0.25)
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.static_loss = step(self.static_batch)
        # the graph's optimiser nodes copy the pinned pointer tables of the capture on every replay: keep a snapshot
        # (and the captured gradient tensors, which live in the graph's private pool) for eager_step()
        self._tables = step.opt.snapshot_pointer_tables()
        self._captured_grads = [p.grad for g in step.opt.param_groups for p in g["params"]]
        # the capture ran the Python side of one step (host step counts advanced) without executing it on the
        # device: bring the device-side count of the captured tick back in line on first replay
        self._first = True
        self._replayed = torch.cuda.Event()

    def __call__(self, batch):
        for dst, src in zip(self.static_batch, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self._first:
            self._first = False      # host counters already include this step (incremented during capture)
        else:
            self.step.opt.note_replayed()
        self.graph.replay()
        self._replayed.record()      # eager_step() waits for it before it touches the optimiser's pointer tables
        # the replay rewrites the parameters through raw pointers: no autograd version bump, no optimiser post-hook.
        # Without this an eager forward between replays (validation) would keep using the split weight images it
        # cached at the previous epoch.
        from . import ops
        ops.weights_updated()
        return self.static_loss

    def eager_step(self, batch):
        """A real (un-captured) optimiser step on ``batch`` - e.g. an odd-shaped last batch of an epoch.  The
        captured graph reads the gradient / moment addresses of ITS step from the optimiser's pinned pointer table;
        an eager step rewrites that table (its gradients are fresh allocations), so the table is restored afterwards
        and the gradients of the capture stay alive (``TrainStep`` keeps them: see ``_captured_grads``).

        ``graph.replay()`` is asynchronous and the captured optimiser node re-reads the pinned table WHEN THE GPU REACHES
        IT: an eager step issued right behind a replay would overwrite the table under a replay that has not got
        there yet (any GPU-bound configuration: ~150 ms of device time against ~30 ms of host enqueue), and that
        replay would apply AdamW with the eager step's gradient addresses.  So the host first waits for the last
        replay to finish (``tests/test_hip_graph.py::test_eager_step_right_behind_a_gpu_bound_replay``)."""
        self._replayed.synchronize()
        loss = self.step(batch)
        self.step.opt.restore_pointer_tables(self._tables)
        return loss


# ---------------------------------------------------------------------------------- data parallel
def init_distributed(backend: Optional[str] = None, capturable: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; no-op for a single process.

    ``capturable``: the DDP step will be captured into a HIP graph (``GraphedTrainStep``): a replay runs its RCCL
    kernels without the watchdog's bookkeeping, so the asynchronous error-handling thread is switched off
    (TORCH_NCCL_ASYNC_ERROR_HANDLING=0).  Eager multi-GPU training keeps PyTorch's default: a failed or hung
    collective on one rank aborts the job instead of hanging it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # PARADIS_DIST_BACKEND=gloo lets a 1-GPU box exercise the N>1 code path (tests only)
            backend = os.environ.get("PARADIS_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            if capturable:
                os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def wrap_ddp(model, *, bucket_cap_mb: float = 32, device_ids: Optional[Sequence[int]] = None,
             static_graph: bool = False, force: bool = False, capturable: bool = False):
    """Batch-sharded data parallelism (the reference's only strategy, ``train.py:49``): full replica
    per GPU, bucketed gradient all-reduce (RCCL over xGMI) overlapped with backward.  Geometry
    buffers are deterministic functions of the grid, so buffer broadcast is disabled; the graph is
    static, gradients alias the buckets.

    ``force``: wrap even in a one-rank process group (a world_size = 1 ``nccl`` group initialises RCCL and runs
    its all-reduce kernels: the one-GPU stand-in for the N > 1 path, tests/test_hip_ddp.py and the ``ddp_overhead_ms``
    leg of bench.py).  ``capturable``: construct the wrapper on a side stream, which is what capturing a DDP step
    into a HIP graph requires (``GraphedTrainStep``; the Reducer's bucket views and streams must not belong to the
    capturing stream)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return model

    def build():
        return torch.nn.parallel.DistributedDataParallel(
            model, device_ids=list(device_ids) if device_ids is not None else None,
            broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb,
            static_graph=bool(static_graph))
    if capturable and torch.cuda.is_available():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ddp = build()
        torch.cuda.current_stream().wait_stream(side)
        return ddp
    return build()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
