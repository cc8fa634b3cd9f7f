"""`python bench.py --gpus N` starts its own ranks (reference train.py:49: one process per GPU, DDP).

CPU: the parent path - it must not need a GPU, must print the launcher command (`--launch-only`), and must refuse a
node with fewer GPUs than asked for with rc 2 (not a traceback from inside a rank).
GPU: the two-rank form end to end on the one GPU of the test box (gloo, both ranks on cuda:0 through the
PARADIS_SHARE_GPU0 hook): one JSON line on stdout, n_gpus = 2, the process group's world size as the ranks saw it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_launch_only_prints_the_rank_command_without_a_gpu():
    r = _run(["--gpus", "4", "--steps", "7", "--warmup", "2", "--launch-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    cmd = rec["launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) == rec["master_port"] > 0
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]     # --launch-only is not passed on
    assert rec["n_gpus"] == 4


def test_bare_multi_gpu_call_without_the_gpus_is_a_clean_refusal():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs")
    r = _run(["--gpus", "2", "--steps", "1"])
    assert r.returncode == 2
    assert r.stdout.strip() == ""
    assert "--gpus 2" in r.stderr and "Traceback" not in r.stderr


def test_inside_a_torchrun_environment_the_launcher_is_not_entered():
    # WORLD_SIZE set = we ARE a rank: --launch-only has nothing to do and says so (no recursion into a second launcher)
    r = _run(["--gpus", "2", "--launch-only"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "nothing to launch" in r.stderr


@pytest.mark.gpu
def test_bare_two_rank_call_runs_end_to_end_on_one_gpu():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-extra-legs", "--no-cpu-baseline",
              "--no-other-configs"],
             env={"PARADIS_SHARE_GPU0": "1", "PARADIS_DIST_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    ddp = rec["config"]["ddp"]
    assert ddp["backend"] == "gloo" and ddp["world_size_seen"] == 2
    assert "self-launch" in ddp["launcher"]
    assert rec["ddp_efficiency_vs"] is None
    assert rec["value"] > 0 and abs(rec["value"] - 4 * 2 / (rec["ms_per_step"] * 2e-3)) < 1e-6 * rec["value"]
