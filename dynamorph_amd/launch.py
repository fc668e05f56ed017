"""One fresh process per GPU for a script that was started as a plain `python script.py --gpus N`.

The reference fans out the same way: `run_VAE.py:10-25, 73-85` starts one `Worker(Process)` per device id with the spawn
start method and joins them.  Here the children are ranks of one `torch.distributed` job (RCCL over xGMI, or gloo for a
rehearsal), started through `torch.distributed.run`, which also sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

The parent never touches the GPU (no HIP call, no torch.cuda call at all: devices are counted from the kernel driver's
topology in /sys), never replaces itself with another program, forwards the ONE JSON line the job's rank 0 prints and
returns the job's exit code.  The job runs in a session of its own: on a time-out, an exception or a SIGTERM / SIGINT to
the parent it is asked to stop (SIGTERM: torchrun's agent tears its ranks down), and after a grace period its whole
process group is killed, so no rank is left holding a GPU.  A failed child is reported, not retried.
"""
import json
import os
import queue
import signal
import socket
import subprocess
import sys
import threading
import time

GRACE_SECONDS = 15.0


def launched():
    """True inside a rank that torch.distributed.run (or this launcher) started."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process' children would see, without initialising HIP: the kfd topology nodes that have SIMDs (CPU nodes
    have none), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists.  None when the
    driver's topology is not there to read (no amdgpu driver: the per-rank check of dist.init_from_env decides then)."""
    try:
        nodes = sorted(os.listdir(topology))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(topology, node, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue
        n += int(props.get("simd_count", "0")) > 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([d for d in v.split(",") if d.strip()]))
    return n


def check_devices(n, device_count=None):
    """One GPU per rank unless DM_DIST_BACKEND=gloo asks for a rehearsal (ranks then share the visible devices).  Every
    rank checks its own device again (dist.init_from_env); this is only the early, readable error."""
    if os.environ.get("DM_DIST_BACKEND") == "gloo":
        return
    if device_count is None:
        device_count = visible_gpu_count()
        if device_count is None:
            return
    if device_count < n:
        raise SystemExit(f"--gpus {n}: {device_count} GPU(s) visible and one rank per GPU is needed for RCCL "
                         f"(DM_DIST_BACKEND=gloo rehearses {n} ranks on fewer devices)")


def _proc_table():
    """{pid: (parent pid, start time)} of every process visible in /proc."""
    table = {}
    for name in os.listdir("/proc"):
        if name.isdigit():
            try:
                tail = open(f"/proc/{name}/stat").read().rsplit(")", 1)[1].split()
                table[int(name)] = (int(tail[1]), tail[19])
            except (OSError, IndexError, ValueError):
                pass
    return table


def _descendants(root):
    """[(pid, start time)] of the processes below `root` (torchrun's ranks run in sessions of their own, so the child's
    process group does not hold them: they are found by parentage, exact pids, never by a pattern)."""
    table = _proc_table()
    kids = {}
    for pid, (ppid, _) in table.items():
        kids.setdefault(ppid, []).append(pid)
    out, todo = [], [root]
    while todo:
        for k in kids.get(todo.pop(), ()):
            out.append((k, table[k][1]))
            todo.append(k)
    return out


def _stop(proc, grace=None):
    """Ends the job this call started: SIGTERM to torchrun (its agent asks the ranks to stop), then after the grace period
    SIGKILL to the child's process group and to every process that was below it -- torchrun gives a rank that ignores
    SIGTERM 30 s and loses it altogether when it is killed itself first, which is how ranks used to be left on a GPU."""
    grace = GRACE_SECONDS if grace is None else grace
    below = _descendants(proc.pid)
    if proc.poll() is None:
        proc.terminate()
        try:
            proc.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            pass
    below = {*below, *_descendants(proc.pid)}
    try:
        os.killpg(proc.pid, signal.SIGKILL)          # (pgid == the child's pid: start_new_session)
    except (ProcessLookupError, PermissionError):
        pass
    now = _proc_table()
    for pid, started in below:
        if pid in now and now[pid][1] == started:    # still the same process, not a reused pid
            try:
                os.kill(pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
    proc.wait()


class _Interrupted(Exception):
    pass


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
    proc = subprocess.Popen(cmd, env=child_env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    lines = queue.Queue()

    def reader():
        for line in proc.stdout:
            lines.put(line)
        lines.put(None)
    threading.Thread(target=reader, daemon=True).start()

    def on_signal(signum, frame):
        raise _Interrupted(signum)
    previous = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGINT):
            previous[sig] = signal.signal(sig, on_signal)
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while True:
            try:
                line = lines.get(timeout=0.5)
            except queue.Empty:
                if deadline is not None and time.monotonic() > deadline:
                    print(f"launch: the {n}-rank job did not finish within {timeout} s: stopping it", file=stderr, flush=True)
                    _stop(proc)
                    return 124
                continue
            if line is None:
                break
            s = line.strip()
            is_json = False
            if s.startswith("{") and s.endswith("}"):
                try:
                    json.loads(s)
                    is_json = True
                except ValueError:
                    pass
            print(s if is_json else line.rstrip("\n"), file=stdout if is_json else stderr, flush=True)
        left = None if deadline is None else max(0.0, deadline - time.monotonic())
        try:
            rc = proc.wait(timeout=left)
        except subprocess.TimeoutExpired:
            print(f"launch: the {n}-rank job did not exit within {timeout} s: stopping it", file=stderr, flush=True)
            _stop(proc)
            return 124
    except _Interrupted as e:
        print(f"launch: signal {e.args[0]}: stopping the {n}-rank job", file=stderr, flush=True)
        _stop(proc)
        return 128 + int(e.args[0])
    except BaseException:
        _stop(proc)
        raise
    finally:
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    try:
        os.killpg(proc.pid, signal.SIGKILL)      # torchrun is gone: nothing of its group may stay behind on a GPU
    except (ProcessLookupError, PermissionError):
        pass
    if rc != 0:
        print(f"launch: the {n}-rank job ended with exit code {rc}", file=stderr, flush=True)
    return rc
