"""bench.py's N > 1 path on a one-GPU box: two rank processes started by bench.py's own launcher, both on cuda:0,
collectives over gloo (`--rehearse-on-one-gpu`; RCCL refuses two ranks on one device).  Checks what a real N = 2 run
must also produce: n_gpus, the oracle check of rank 0's batch, the key broadcast and the sharded config-5 leg with its
own oracle check.  The numbers of such a run are not measurements."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_keep_parity_and_shard_config5():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--skip-ntt",
           "--steps", "2", "--warmup", "1", "--batch", "2", "--config5-pairs", "33", "--config5-slice", "16"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["rehearsal"]
    assert d["parity_checked"] is True
    assert d["config"]["key_broadcast_s"] > 0
    c5 = d["config5"]
    assert c5["pairs"] == 33 and c5["pairs_rank0"] == 17 and c5["n_gpus"] == 2 and c5["scaling"] == "strong"
    assert c5["parity_checked"] is True
