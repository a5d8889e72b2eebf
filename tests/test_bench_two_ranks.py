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
