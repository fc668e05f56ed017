"""`not gpu`: `python bench.py --gpus N` started plainly must start its own N ranks (dynamorph_amd/launch.py; the
reference fans out one Process per device, run_VAE.py:10-25, 73-85).  The launcher is exercised here with a stand-in rank
script on gloo / CPU; bench.py's own use of it is checked by reading its start-up path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "helpers", "launch_standin.py")


def _run(*argv, env_extra=None):
    env = dict(os.environ, DM_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, STANDIN, *argv], env=env, capture_output=True, text=True, timeout=300)


def test_plain_invocation_starts_its_own_ranks_and_forwards_one_json_line():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # the JSON line alone: the chatter went to stderr
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 2, "sum": 3.0, "max_rank": 1.0, "self_launched": "1"}
    assert "chatter on stdout" in r.stderr and "launch:" in r.stderr


def test_a_failed_rank_is_reported_by_the_exit_code_not_retried():
    r = _run("--gpus", "2", "--fail-rank", "1")
    assert r.returncode != 0
    assert r.stderr.count("launch: ") == 2                 # the command line once, the verdict once: no second attempt
    assert "exit code" in r.stderr


def test_single_process_line_is_unchanged():
    r = _run("--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["self_launched"] is None


def test_missing_gpus_are_a_clear_error_unless_gloo_is_asked_for():
    from dynamorph_amd import launch
    os.environ.pop("DM_DIST_BACKEND", None)
    with pytest.raises(SystemExit) as e:
        launch.check_devices(8, device_count=1)
    assert "--gpus 8" in str(e.value) and "gloo" in str(e.value)
    launch.check_devices(1, device_count=1)
    os.environ["DM_DIST_BACKEND"] = "gloo"
    try:
        launch.check_devices(8, device_count=1)
    finally:
        os.environ.pop("DM_DIST_BACKEND", None)


def test_bench_takes_the_launcher_before_anything_touches_the_gpu():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    launch_at = main.index("launch.self_launch(")
    for gpu_call in ("D.init_from_env(", "torch.cuda.set_device(", ".to(dev)", "is_available("):
        at = main.find(gpu_call)
        assert at < 0 or at > launch_at, gpu_call
    assert "os.exec" not in src and "execv" not in src
