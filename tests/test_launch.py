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


def test_eight_ranks_like_the_8_gpu_node_of_config_c4():
    """C4's launch shape without the hardware: 8 ranks (gloo / CPU) through the same launcher and the same dist helpers
    bench.py's N > 1 line is built from -- world 8, one time per rank, and the gradient weights of a ragged global batch
    (13 samples over 8 ranks: five shards of 2, three of 1) that make mean-over-ranks the global-batch mean."""
    r = _run("--gpus", "8", "--ragged-batch", "13")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["sum"] == 36.0 and rec["max_rank"] == 7.0
    assert rec["collective"]["world"] == 8 and rec["collective"]["backend"] == "gloo"
    assert rec["collective"]["rank_ms_per_step"] == [1.0 + k for k in range(8)]
    w = rec["shard_weights"]
    assert w == [2 * 8 / 13] * 5 + [1 * 8 / 13] * 3 and abs(sum(w) - 8.0) < 1e-12


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                                    # (a zombie still answers kill -0)
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _rank_pids(pid_dir, n, wait_s=120):
    import time
    t0 = time.time()
    while time.time() - t0 < wait_s:
        have = [f for f in os.listdir(pid_dir) if f.endswith(".pid")]
        if len(have) == n and all(open(os.path.join(pid_dir, f)).read().strip() for f in have):
            return [int(open(os.path.join(pid_dir, f)).read()) for f in have]
        time.sleep(0.2)
    raise AssertionError(f"{n} ranks did not start within {wait_s} s")


def test_a_hung_rank_is_stopped_at_the_timeout_and_no_rank_survives(tmp_path):
    """A rank that never finishes and ignores SIGTERM: the launcher's timeout holds (it used to sit in the read loop until
    EOF), the job's own process group is killed, exit code 124."""
    import time
    r = _run("--gpus", "2", "--hang-rank", "1", "--launch-timeout", "25", "--pid-dir", str(tmp_path))
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "did not finish within" in r.stderr or "did not exit within" in r.stderr
    pids = _rank_pids(str(tmp_path), 2, wait_s=1)
    time.sleep(0.5)
    assert not any(_alive(p) for p in pids), "a rank outlived the launcher"


def test_sigterm_to_the_parent_takes_the_whole_job_down(tmp_path):
    """A harness that SIGTERMs `python bench.py --gpus N` must not leave torchrun and its ranks on the GPUs."""
    import signal
    import time
    env = dict(os.environ, DM_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, STANDIN, "--gpus", "2", "--hang-rank", "0", "--pid-dir", str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        pids = _rank_pids(str(tmp_path), 2)
        p.send_signal(signal.SIGTERM)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    assert "stopping the 2-rank job" in err
    time.sleep(0.5)
    assert not any(_alive(q) for q in pids), "a rank outlived the launcher"


def test_gpus_are_counted_without_touching_hip(tmp_path, monkeypatch):
    """launch.visible_gpu_count reads the kfd topology (nodes with SIMDs) and the *_VISIBLE_DEVICES lists; the parent of a
    multi-rank job makes no torch.cuda call at all."""
    from dynamorph_amd import launch
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):     # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert launch.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert launch.visible_gpu_count(str(tmp_path)) == 2
    assert launch.visible_gpu_count(str(tmp_path / "missing")) is None
    src = open(os.path.join(ROOT, "dynamorph_amd", "launch.py")).read()
    assert "import torch" not in src and "torch.cuda" not in src.split('"""', 2)[2]


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
