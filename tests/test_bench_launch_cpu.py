"""`python bench.py --gpus N` must launch its own ranks: run the N = 2 launcher path on CPU with the
stubbed loop (gloo) and check the single rank-0 JSON line."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_prints_one_json_line():
    out = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-cpu"])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["scaling"] == "weak"
    # max over ranks: rank 1 sleeps 2x as long as rank 0
    assert out["ms_per_step"] >= 2 * 2.0 * 0.9
    for key in ("metric", "value", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out


def test_single_rank_needs_no_launcher():
    out = _run(["--steps", "3", "--stub-cpu"])
    assert out["n_gpus"] == 1


def test_rank_without_a_peer_fails_loud_with_a_json_error_line():
    """torchrun-style environment for rank 1 of 2 with nobody hosting the rendezvous: exit code 3 and ONE JSON line naming the cause."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TT_BENCH_RENDEZVOUS_TIMEOUT_S="3")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--stub-cpu"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    err = json.loads(lines[0])["error"]
    assert "rank 1" in err and "init_process_group" in err
