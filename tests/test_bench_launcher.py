"""bench.py --gpus N without an external launcher: the parent starts N rank processes of itself before anything
touches the GPU and relays rank 0's line.  Driven here with world size 2 over gloo on the
CPU through `--launcher-selftest`, which runs the rendezvous, the barrier, the max-over-ranks reduction and the
config-5 shard arithmetic of the real bench without any GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def test_gpus_2_spawns_two_ranks_and_relays_rank0():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["selftest"] is True
    assert rec["max_over_ranks"] == 2.0  # rank 1 reported 2.0, rank 0 1.0
    assert rec["config5_pairs"] == 8192  # the two shard_range slices cover the whole batch


def test_gpus_must_match_the_external_launcher():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2
    assert "error" in json.loads(r.stdout.strip().splitlines()[-1])


def test_a_failing_rank_fails_the_launcher():
    # rank 1 exits with 3 before the rendezvous; the launcher must stop rank 0 (left waiting) and report the code
    env = _env()
    env["CAT_BENCH_SELFTEST_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 3
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_under_an_external_launcher_gpus_defaults_to_world_size():
    # `torchrun --nproc-per-node 2 bench.py` without --gpus: the launcher's WORLD_SIZE is adopted
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--launcher-selftest"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 2
