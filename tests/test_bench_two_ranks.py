"""N = 2 control flow of bench.py on ONE GPU: two ranks launched exactly as the driver does (torch.distributed.run, one
process per rank) sharing cuda:0 with the gloo backend (RCCL refuses two ranks on one device; the collectives, their order,
the overlapped buckets and the timing protocol are the same code).  Checks the single JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu_gloo():
    env = dict(os.environ, GDMAE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--config", "A", "--batch-per-gpu", "2", "--mask-ratio", "0.5", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["global_batch"] == 4
    gs = out["grad_sync"]
    assert [b for b, _ in gs["buckets"]] == ["vfe", "backbone_3d.sst_blocks.0", "backbone_3d.decoder"]
    assert gs["last_step"] == [["backbone_3d.decoder", "overlapped"], ["backbone_3d.sst_blocks.0", "overlapped"], ["vfe", "tail"]]


@pytest.mark.gpu
def test_bench_config_c_shape_over_rccl_one_rank():
    """Rehearsal of the multi-GPU bench on the one GPU of the test box, over RCCL itself (a one-rank nccl group with the gradient
    exchange forced on): the config-C batch (4 full-size frames per GPU), overlapped buckets from inside the backward, barrier +
    MAX-reduce timing protocol, teardown.  What the 8-GPU run adds is ranks, not code."""
    env = dict(os.environ, GDMAE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3", "--batch-per-gpu", "4",
           "--no-cpu-baseline", "--no-also", "--no-roofline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["frames_per_gpu"] == 4 and out["value"] > 0
    gs = out["grad_sync"]
    names = [b for b, _ in gs["buckets"]]
    assert names[0] == "vfe" and names[-1] == "backbone_3d.decoder" and len(names) == 5
    assert gs["last_step"] == [[names[-1], "overlapped"]] + [[b, "overlapped"] for b in reversed(names[1:-1])] + [["vfe", "tail"]]
