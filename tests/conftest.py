import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "cal_22-mpc_amd"

# The HIP runtime says nothing at its default log level before it gives up on a fatal error (one GPU test run of round 3 ended in
# a bare abort() inside mpc_compress_batch, not reproducible, with an empty stderr): errors only, set before the runtime starts.
os.environ.setdefault("AMD_LOG_LEVEL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    """Import the package (its directory name has a dash, so go through importlib)."""
    return importlib.import_module(PKG if sub is None else f"{PKG}.{sub}")


@pytest.fixture(scope="session")
def configs():
    return pkg("configs")


@pytest.fixture(scope="session")
def traces():
    return pkg("traces")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(ref=True)
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- a clean box after the session ---------------------------------------------------------------------------------
# Several tests start child processes (torch.distributed.run, multiprocessing spawn workers, the command line
# binary, fresh interpreters for the grid-capped runs).  None of them may outlive the session, and nothing may be left
# in /dev/shm (shared-memory segments live in RAM): the session ends by checking both, stopping what it finds (exact
# process ids only) and failing if a worker of ours was still alive.
def _descendants(root_pid):
    kids = {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open(f"/proc/{name}/stat") as f:
                st = f.read()
            rest = st[st.rindex(")") + 2:].split()
            kids.setdefault(int(rest[1]), []).append((int(name), st[st.index("(") + 1:st.rindex(")")], rest[0]))
        except (OSError, ValueError):
            continue
    out, todo = [], [root_pid]
    while todo:
        for pid, comm, state in kids.get(todo.pop(), []):
            out.append((pid, comm, state))
            todo.append(pid)
    return out


def _shm_names():
    try:
        return set(os.listdir("/dev/shm"))
    except OSError:
        return set()


def pytest_sessionstart(session):
    session.config._mpc_shm_before = _shm_names()
    # Code objects of module sequences compiled at handle creation (csrc/mpc_jit.h) go to a cache directory; the default is
    # under $HOME and would stay behind on the box.  The session (and the child processes its tests start) gets one of its
    # own, removed when the session ends.
    if "MPC_JIT_CACHE" not in os.environ:
        import tempfile
        session.config._mpc_jit_cache = tempfile.mkdtemp(prefix="mpc_jit_cache_")
        os.environ["MPC_JIT_CACHE"] = session.config._mpc_jit_cache
    # Orphans come home.  A helper whose parent exits first (the resource tracker of a torch.distributed.run agent, for
    # one) is re-parented to process 1, which on the GPU boxes never reaps: it stays behind as a zombie after the
    # session (seen in round 2: "the box could not be shown clean").  As a child subreaper this process inherits such
    # orphans instead and reaps them at the end of the session.
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(36, 1, 0, 0, 0)      # PR_SET_CHILD_SUBREAPER
    except Exception:
        pass


def _reap_children():
    n = 0
    while True:
        try:
            pid, _ = os.waitpid(-1, os.WNOHANG)
        except ChildProcessError:
            break
        if pid == 0:
            break
        n += 1
    return n


def _stop_resource_tracker():
    """multiprocessing's resource tracker is a helper child of THIS process (started by the first spawn); it only exits
    when this process does and would then be an orphan of process 1 for a while -- what the round-2 record saw after
    the GPU test step.  Stop it (close its pipe, wait for it) while we are still its parent."""
    try:
        from multiprocessing import resource_tracker
        rt = getattr(resource_tracker, "_resource_tracker", None)
        if rt is not None and getattr(rt, "_pid", None) is not None and hasattr(rt, "_stop"):
            rt._stop()
    except Exception:
        pass


def pytest_sessionfinish(session, exitstatus):
    import shutil
    import signal
    import time
    me = os.getpid()
    jit_cache = getattr(session.config, "_mpc_jit_cache", None)
    if jit_cache:
        shutil.rmtree(jit_cache, ignore_errors=True)
        os.environ.pop("MPC_JIT_CACHE", None)
    _stop_resource_tracker()
    # helper daemons of multiprocessing / torch end with this process; anything else is a worker that leaked
    benign = ("resource_tracker", "torch_shm_manag")
    _reap_children()
    left = [(pid, comm, state) for pid, comm, state in _descendants(me) if state != "Z"]
    workers = []
    for pid, comm, state in left:
        try:
            with open(f"/proc/{pid}/cmdline", "rb") as f:
                cmd = f.read().replace(b"\0", b" ").decode(errors="replace")
        except OSError:
            continue
        if any(b in comm or b in cmd for b in benign):
            continue
        workers.append((pid, comm, cmd[:200]))
    for pid, _, _ in workers:
        try:
            os.kill(pid, signal.SIGTERM)
        except OSError:
            pass
    if workers:
        time.sleep(1.0)
        for pid, _, _ in workers:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
    time.sleep(0.2)
    _reap_children()           # exited children and adopted orphans: no zombies stay behind
    new_shm = sorted(_shm_names() - getattr(session.config, "_mpc_shm_before", set()))
    for name in new_shm:
        try:
            os.unlink(os.path.join("/dev/shm", name))
        except OSError:
            pass
    if workers or new_shm:
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        msg = f"session left behind: processes {workers} /dev/shm {new_shm}"
        if tr:
            tr.write_line(msg, red=True)
        else:
            print(msg)
    if workers:
        session.exitstatus = 1
