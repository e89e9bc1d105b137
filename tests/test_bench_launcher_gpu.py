"""bench.py under the launcher the driver uses for N > 1 (`python -m torch.distributed.run`), with one rank:
the RANK / LOCAL_RANK / WORLD_SIZE plumbing, the RCCL process group bound to the rank's device, the barrier
and the MAX all-reduce of the elapsed time all execute on the GPU, and the JSON line keeps its contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_under_torch_distributed_run_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-extra-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["unit"] == "MP/s" and rec["value"] > 0 and rec["scaling"] == "weak"
    assert rec["config"]["batch_per_gpu"] == 2


def test_training_step_bench_runs_under_the_launcher_with_one_rank():
    """`bench.py --train-step` the way a multi-GPU run starts it.  The process group exists, so every step runs the
    all-reduce of the flat gradient buffer (26.3 M floats, 105 MB) through RCCL on the GPU - over one rank here, the
    same call a node of eight makes - and the bench times that collective alone as `allreduce_ms`.  (The parameter
    broadcast and the loss mean are skipped for a single rank: they have nothing to send.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29618", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--train-step", "--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["unit"] == "MP/s" and rec["value"] > 0
    assert rec["config"]["batch_per_gpu"] == 2 and rec["config"]["gradient_floats"] > 26_000_000
    assert 0 < rec["loss_last"] < 2 * rec["loss_first"]
    assert rec["allreduce_ms"] > 0 and rec["allreduce_floats"] == rec["config"]["gradient_floats"]


def test_images_mode_runs_under_the_launcher_with_one_rank():
    """`bench.py --images`: a folder of Rain100L-sized images dealt over the ranks by harness.restore_images and
    gathered on rank 0 (BASELINE configs[2]); with one rank the deal, the per-image batches and the result order
    still execute."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--images", "6", "--arch", "ASTv2", "--batch", "4", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["images"] == 6 and rec["config"]["height"] == 321 and rec["config"]["width"] == 481
