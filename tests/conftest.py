import faulthandler
import os
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# ---- hang instrumentation (one full `-m gpu` run of r04 sat in one test until the 40-minute limit of its call) -------------
# (1) every test appends its node id to a progress file when it starts and when it ends: a run that is cut off names the test it was in;
# (2) a watchdog thread looks at the running test every few seconds; after WATCHDOG_S it writes, ONCE per test, the Python stacks of all threads
#     (faulthandler: works while the main thread sits in a C call), the library's own account of its last launches (fvsrn_debug_state: kernel,
#     launch shape, busy stream, device work counters -- never blocks) and the child processes of this one (the 2-rank tests start rank processes);
# (3) GPU tests time out with pytest-timeout's THREAD method: the signal method cannot interrupt a main thread that is blocked inside
#     hipStreamSynchronize / a ctypes call, which is exactly how a hung kernel or a lost rendezvous looks from here.
PROGRESS = os.environ.get("FVSRN_TEST_PROGRESS") or os.path.join(ROOT, "gpurun_out", "pytest_progress_%d.log" % os.getpid())
WATCHDOG_S = float(os.environ.get("FVSRN_TEST_WATCHDOG_S", "300"))
_state = {"node": None, "t0": 0.0, "dumped": False, "file": None}


def _progress(line):
    try:
        if _state["file"] is None:
            os.makedirs(os.path.dirname(PROGRESS), exist_ok=True)
            _state["file"] = open(PROGRESS, "a", buffering=1)
        _state["file"].write("%.3f %s\n" % (time.time(), line))
    except OSError:
        _state["file"] = False  # (read-only tree: run without the file)


def _dump(node, elapsed):
    f = _state["file"] or sys.stderr
    for out in {f, sys.stderr}:
        try:
            out.write("\n==== watchdog: %s has been running for %.0f s ====\n" % (node, elapsed))
            faulthandler.dump_traceback(file=out, all_threads=True)
            if "fvsrn_amd.capi" in sys.modules:  # (only if the test loaded the library: the watchdog must not)
                out.write("---- fvsrn_debug_state ----\n%s" % sys.modules["fvsrn_amd.capi"].debug_state())
            try:
                import psutil
                for c in psutil.Process().children(recursive=True):
                    out.write("child %d %s cpu %.1f s: %s\n" % (c.pid, c.status(), sum(c.cpu_times()[:2]), " ".join(c.cmdline())[:300]))
            except Exception as e:  # noqa: BLE001
                out.write("children: %r\n" % (e,))
            out.flush()
        except Exception as e:  # noqa: BLE001  (a diagnostic must not take the run down)
            sys.stderr.write("watchdog dump failed: %r\n" % (e,))


def _watchdog():
    while True:
        time.sleep(5.0)
        node, t0 = _state["node"], _state["t0"]
        if node and not _state["dumped"] and time.time() - t0 > WATCHDOG_S:
            _state["dumped"] = True
            _dump(node, time.time() - t0)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    threading.Thread(target=_watchdog, name="fvsrn-test-watchdog", daemon=True).start()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    _state.update(node=item.nodeid, t0=time.time(), dumped=False)
    _progress("START " + item.nodeid)
    yield
    _progress("END   %s %.2f s" % (item.nodeid, time.time() - _state["t0"]))
    _state["node"] = None


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A hung GPU test must fail, not stall the suite: 900 s per test; method "thread" for GPU tests (see (3) above: it ends the whole run with the stacks
    # of all threads on stderr -- a suite that stops at the hang with a diagnosis beats one that sits in it).  pytest-timeout is part of the image;
    # without it the suite runs unguarded.
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(900, method="thread") if "gpu" in item.keywords else pytest.mark.timeout(900))
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
