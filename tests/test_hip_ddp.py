"""GPU: the data-parallel path with the real HIP model.  Only one GPU is available to the test
box, so the two ranks share cuda:0 and use the gloo backend (RCCL refuses two ranks on one device);
what is exercised is DistributedDataParallel + the custom autograd ops (bucket views, hooks,
overlap with backward), which is backend independent."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, static_graph=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from paradis_model_amd.config import reduced_config, stub_datamodule
    from paradis_model_amd.harness import TrainStep, init_distributed, make_grids, synthetic_batch, wrap_ddp
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = reduced_config()
    lat_deg, lg, og = make_grids(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    ddp = wrap_ddp(model, bucket_cap_mb=0.1, device_ids=[0], static_graph=static_graph)
    # Overlap evidence: a communication hook sees every bucket's all-reduce being launched; autograd hooks on
    # the parameters count how many gradients had NOT been produced yet at that moment.  Buckets that fire while
    # gradients are still outstanding are collectives running under the rest of the backward pass.
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    pending = {"n": 0}
    fired = []
    n_params = sum(1 for p in model.parameters() if p.requires_grad)
    for p in model.parameters():
        if p.requires_grad:
            p.register_post_accumulate_grad_hook(lambda _p: pending.__setitem__("n", pending["n"] - 1))

    def hook(state, bucket):
        fired.append((bucket.index(), pending["n"]))
        return default_hooks.allreduce_hook(state, bucket)

    ddp.register_comm_hook(dist.group.WORLD, hook)
    step = TrainStep(ddp, build_loss(cfg, lat_deg).to(dev), cfg)
    full = synthetic_batch(16, 32, False, 2 * world, 1, seed=5, device=dev)
    shard = tuple(t[rank * 2:(rank + 1) * 2] for t in full)
    losses = []
    for _ in range(2):
        pending["n"] = n_params
        fired.clear()
        losses.append(float(step(shard)))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    torch.save({"params": flat, "losses": losses, "fired": list(fired), "n_params": n_params},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_ddp_with_hip_model_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["params"], r1["params"])
    # overlap exists: at least two buckets' all-reduces were launched while gradients were still outstanding
    # (0.1 MB buckets over the reduced model's 0.47 MB of gradients), and the last bucket fired when none were
    fired = r0["fired"]
    assert len(fired) >= 3, fired
    early = [b for b, left in fired if left > 0]
    assert len(early) >= 2, fired
    assert min(left for _, left in fired) == 0, fired
    # static_graph=True: bit-identical parameters after the same two steps
    sg = tmp_path / "sg"
    sg.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(sg), True), nprocs=world, join=True)
    s0 = torch.load(sg / "rank0.pt")
    assert torch.equal(s0["params"], r0["params"]) and s0["losses"] == r0["losses"]
    sys.path.insert(0, ROOT)
    from paradis_model_amd.config import reduced_config, stub_datamodule
    from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    dev = torch.device("cuda", 0)
    cfg = reduced_config()
    lat_deg, lg, og = make_grids(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    step = TrainStep(model, build_loss(cfg, lat_deg).to(dev), cfg)
    batch = synthetic_batch(16, 32, False, 2 * world, 1, seed=5, device=dev)
    losses = [float(step(batch)) for _ in range(2)]
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    for a, b0, b1 in zip(losses, r0["losses"], r1["losses"]):
        assert abs(a - 0.5 * (b0 + b1)) < 1e-5 * abs(a)
    assert float((flat - r0["params"]).abs().max()) < 5e-4
    assert float((flat - r0["params"]).abs().mean()) < 2e-5


def _nccl_worker(rank, port, out_dir, graphed):
    """ONE rank on cuda:0 with the nccl (= RCCL) backend: the process group really initialises RCCL and the Reducer's
    bucket all-reduces run as RCCL kernels - what the gloo tests above cannot show on a one-GPU box."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from paradis_model_amd.config import reduced_config, stub_datamodule
    from paradis_model_amd.harness import GraphedTrainStep, TrainStep, make_grids, synthetic_batch, wrap_ddp
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    cfg = reduced_config()
    lat_deg, lg, og = make_grids(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    ddp = wrap_ddp(model, bucket_cap_mb=0.1, device_ids=[0], force=True, capturable=graphed)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    step = TrainStep(ddp, build_loss(cfg, lat_deg).to(dev), cfg, capturable=graphed)
    batches = [synthetic_batch(16, 32, False, 2, 1, seed=5 + i, device=dev) for i in range(2)]
    n_steps = 14
    order = [0] * GraphedTrainStep.DDP_WARMUP + [i % 2 for i in range(n_steps - GraphedTrainStep.DDP_WARMUP)]
    rec = {}
    if graphed:
        g = GraphedTrainStep(step, batches[0], warmup=2)       # raised to DDP_WARMUP by the class
        assert g.warmup_steps == GraphedTrainStep.DDP_WARMUP
        losses = [float(g(batches[i])) for i in order[GraphedTrainStep.DDP_WARMUP:]]
    else:
        losses = [float(step(batches[i])) for i in order]
        # gradient_as_bucket_view with the custom autograd ops: parameter gradients are views into the Reducer's
        # flat buckets - several gradients share one storage that is larger than any of them
        by_storage = {}
        for p in model.parameters():
            if p.grad is not None:
                by_storage.setdefault(p.grad.untyped_storage().data_ptr(), []).append(p)
        shared = [ps for ps in by_storage.values() if len(ps) >= 2]
        rec["n_grads"] = sum(len(ps) for ps in by_storage.values())
        rec["n_in_shared_storage"] = sum(len(ps) for ps in shared)
        rec["view_storage_larger"] = all(ps[0].grad.untyped_storage().nbytes() >= sum(q.grad.numel() for q in ps) * 4
                                         for ps in shared)
    torch.cuda.synchronize()
    rec.update(params=torch.cat([p.detach().flatten() for p in model.parameters()]).cpu(), losses=losses)
    torch.save(rec, os.path.join(out_dir, f"nccl_{int(graphed)}.pt"))
    # ordered teardown: a captured graph holds RCCL nodes - it goes before the communicator does
    if graphed:
        del g
    del step, ddp
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_one_rank_nccl_ddp_eager_and_graphed_match_plain(tmp_path):
    """RCCL through a world_size = 1 nccl group: (i) the eager DDP step equals the plain step and its gradients alias
    the buckets; (ii) the DDP step captured into ONE HIP graph - forward, loss, backward with the bucket all-reduces
    on the process group's stream, AdamW - replays to the same parameters (reference train.py:47-49: the metric's
    configuration is DDP)."""
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path), False), nprocs=1, join=True)
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path), True), nprocs=1, join=True)
    e, g = torch.load(tmp_path / "nccl_0.pt"), torch.load(tmp_path / "nccl_1.pt")
    assert e["n_in_shared_storage"] >= 0.5 * e["n_grads"] and e["view_storage_larger"], (e["n_grads"], e["n_in_shared_storage"])
    # plain twin in this process
    sys.path.insert(0, ROOT)
    from paradis_model_amd.config import reduced_config, stub_datamodule
    from paradis_model_amd.harness import GraphedTrainStep, TrainStep, make_grids, synthetic_batch
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    dev = torch.device("cuda", 0)
    cfg = reduced_config()
    lat_deg, lg, og = make_grids(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).to(dev)
    step = TrainStep(model, build_loss(cfg, lat_deg).to(dev), cfg)
    batches = [synthetic_batch(16, 32, False, 2, 1, seed=5 + i, device=dev) for i in range(2)]
    W = GraphedTrainStep.DDP_WARMUP
    order = [0] * W + [i % 2 for i in range(14 - W)]
    losses = [float(step(batches[i])) for i in order]
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    from tests._util import max_rel
    assert max_rel(e["params"], flat) <= 1e-6, max_rel(e["params"], flat)
    assert max_rel(g["params"], flat) <= 1e-6, max_rel(g["params"], flat)
    for a, b in zip(losses[W:], g["losses"]):
        assert abs(a - b) <= 1e-6 * abs(a), (losses, g["losses"])
