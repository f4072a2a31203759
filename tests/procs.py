"""How tests start other processes that use the GPU (rank processes, bench.py, native checkers): own process group, killed with
the group in a `finally`, killed by the kernel if pytest itself dies (PR_SET_PDEATHSIG), stdout / stderr kept in files whose tail is
shown on failure.  Never re-execs anything: fresh children only."""
import ctypes
import os
import signal
import subprocess
import time

# ------------------------------------------------------------------------------------------------------------------------
_libc = ctypes.CDLL("libc.so.6", use_errno=True)   # loaded HERE, in the parent: a forked child of a many-threaded process (pytest with torch) must not dlopen / import anything


def _die_with_parent():
    """preexec: SIGKILL this child when the process that started it dies (Linux PR_SET_PDEATHSIG = 1), so an aborted pytest
    leaves no rank on the GPU.  Nothing but the one foreign call: no import, no allocation worth the name."""
    _libc.prctl(1, signal.SIGKILL, 0, 0, 0)


def _kill_group(p):
    if p.poll() is None:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    try:
        p.wait(timeout=10)
    except Exception:
        pass


def _tail(path, n=3000):
    try:
        with open(path, "rb") as f:
            b = f.read()
        return b[-n:].decode(errors="replace")
    except OSError:
        return ""


class Spawned:
    def __init__(self, returncode, stdout, stderr_tail):
        self.returncode, self.stdout, self.stderr_tail = returncode, stdout, stderr_tail


def run_ranks(cmds_envs, log_dir, timeout=600):
    """Start one process per (argv, env) — each in its own process group, stderr to <log_dir>/proc<i>.err, stdout to proc<i>.out — wait
    for all of them, and ALWAYS leave none behind.  Returns [Spawned]; a time-out kills everything and raises with the stderr tails."""
    os.makedirs(str(log_dir), exist_ok=True)
    procs, files = [], []
    try:
        for i, (argv, env) in enumerate(cmds_envs):
            fo = open(os.path.join(str(log_dir), f"proc{i}.out"), "wb")
            fe = open(os.path.join(str(log_dir), f"proc{i}.err"), "wb")
            files += [fo, fe]
            procs.append(subprocess.Popen(argv, env=env, stdout=fo, stderr=fe, start_new_session=True, preexec_fn=_die_with_parent))
        end = time.time() + timeout
        for p in procs:
            try:
                p.wait(timeout=max(0.1, end - time.time()))
            except subprocess.TimeoutExpired:
                tails = "\n".join(f"--- proc{i} stderr tail\n{_tail(os.path.join(str(log_dir), f'proc{i}.err'))}" for i in range(len(procs)))
                raise AssertionError(f"rank processes still running after {timeout}s\n{tails}")
    finally:
        for p in procs:
            _kill_group(p)
        for f in files:
            f.close()
    out = []
    for i, p in enumerate(procs):
        with open(os.path.join(str(log_dir), f"proc{i}.out"), "rb") as f:
            so = f.read().decode(errors="replace")
        out.append(Spawned(p.returncode, so, _tail(os.path.join(str(log_dir), f"proc{i}.err"))))
    return out


def spawn(argv, env, log_dir, timeout=600):
    """One child process under the same rules; returns Spawned."""
    return run_ranks([(argv, env)], log_dir, timeout)[0]


def describe(results):
    """Failure text: return codes + every process's stderr tail."""
    return "\n".join(f"--- proc{i} rc={r.returncode}\n{r.stderr_tail}" for i, r in enumerate(results))
