"""One fresh process per GPU for a script that was started as a plain `python script.py --gpus N`.

The reference fans out the same way: `run_VAE.py:10-25, 73-85` starts one `Worker(Process)` per device id with the spawn
start method and joins them.  Here the children are ranks of one `torch.distributed` job (RCCL over xGMI, or gloo for a
rehearsal), started through `torch.distributed.run`, which also sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

The parent never touches the GPU (no HIP call, no `torch.cuda.is_available()`; `torch.cuda.device_count()` does not
initialise the runtime on this image), never replaces itself with another program, forwards the ONE JSON line the job's
rank 0 prints and returns the job's exit code.  A failed child is reported, not retried.
"""
import json
import os
import socket
import subprocess
import sys


def launched():
    """True inside a rank that torch.distributed.run (or this launcher) started."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_devices(n, device_count=None):
    """One GPU per rank unless DM_DIST_BACKEND=gloo asks for a rehearsal (ranks then share the visible devices)."""
    if os.environ.get("DM_DIST_BACKEND") == "gloo":
        return
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()
    if device_count < n:
        raise SystemExit(f"--gpus {n}: {device_count} GPU(s) visible and one rank per GPU is needed for RCCL "
                         f"(DM_DIST_BACKEND=gloo rehearses {n} ranks on fewer devices)")


def self_launch(script, argv, n, timeout=None, env=None, stdout=None, stderr=None):
    """Runs `script argv...` as n ranks on this node and returns the job's exit code.

    stdout of the job is read line by line: JSON objects (the bench line) go to `stdout`, everything else (library
    chatter that happens to use stdout) goes to `stderr`, so that the caller's stdout carries the JSON line alone."""
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what this pool's driver supports (RCCL needs it)
    child_env["DM_SELF_LAUNCHED"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        child_env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    print("launch:", " ".join(cmd), file=stderr, flush=True)
    proc = subprocess.Popen(cmd, env=child_env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            s = line.strip()
            is_json = False
            if s.startswith("{") and s.endswith("}"):
                try:
                    json.loads(s)
                    is_json = True
                except ValueError:
                    pass
            print(s if is_json else line.rstrip("\n"), file=stdout if is_json else stderr, flush=True)
        rc = proc.wait(timeout=timeout)
    except BaseException:
        proc.kill()          # exactly the process this call started (torchrun ends its ranks with it)
        proc.wait()
        raise
    if rc != 0:
        print(f"launch: the {n}-rank job ended with exit code {rc}", file=stderr, flush=True)
    return rc
