import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """The oracle side of the parity tests runs torch / BLAS on the host: a test box gives the process 16 cores of a 256-thread
    host, and a thread per hardware thread spends its time switching (the AUC test: 320 s -> 140 s per dtype)."""
    import os
    import torch
    n = len(os.sched_getaffinity(0))
    try:                                   # the container's CPU quota (cgroup v2)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        n = min(n, 16)
    torch.set_num_threads(max(1, n))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_logstart(nodeid, location):
    if os.environ.get("GRAFT_REPO_ROOT") or os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        heartbeat(f"start {nodeid}")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mindrec_amd import _lib
    rc = _lib.lib().mrec_device_ok()
    assert rc == 0, f"libmrec_hip.so cannot use this device (rc={rc})"
    return torch.device("cuda:0")


def heartbeat(msg):
    """A long test's progress line, appended to gpurun_out/test_progress.log (pytest captures stdout; the GPU box's watchdog
    takes a run that writes nothing for 7 minutes to be hung)."""
    import time
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_progress.log"), "a") as f:
            f.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
    except OSError:
        pass
