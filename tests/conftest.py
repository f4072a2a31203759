"""Suite plumbing.  Three jobs, all about keeping a crash ATTRIBUTABLE and contained (round 5's driver run died with SIGABRT in
test #132 and the only thing the 2 KB log tail held was a faulthandler dump):
  * every test's nodeid is written (flushed) to the terminal BEFORE it runs and its outcome after, so the tail of an aborted log
    names the running test (pytest.ini turns the faulthandler dump off);
  * tests that start other GPU processes are collected LAST (marker `multiprocess`, eight-rank cases after everything else), so the
    parity tests of every SURVEY.md section 8 row have reported before the most fragile tests start;
  * tests/procs.py (`spawn()` / `run_ranks()`) is the one way tests start rank processes — own process group, killed with the group in a `finally`,
    killed by the kernel if pytest itself dies (PR_SET_PDEATHSIG), stderr kept in a file whose tail is shown on failure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multiprocess: starts other processes that use the GPU; collected after the single-process tests")
    config.addinivalue_line("markers", "manyranks: more than two rank processes on one GPU; collected last of all")


def _rank_of(item):
    if item.get_closest_marker("manyranks"):
        return 2
    if item.get_closest_marker("multiprocess"):
        return 1
    return 0


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank_of)          # stable: file order kept inside each class


def _emit(config, text):
    tr = config.pluginmanager.get_plugin("terminalreporter")
    if tr is None:
        return
    tr.ensure_newline()
    tr.write_line(text)
    try:
        tr._tw.flush()
    except Exception:
        pass
    try:
        sys.__stdout__.flush()
    except Exception:
        pass


_t0 = {}


def pytest_runtest_logstart(nodeid, location):
    _t0[nodeid] = time.time()
    if _cfg[0] is not None and _cfg[0].getoption("verbose") <= 0:       # -v already prints the nodeid
        _emit(_cfg[0], f"RUN  {nodeid}")


def pytest_runtest_logfinish(nodeid, location):
    if _cfg[0] is not None and _cfg[0].getoption("verbose") <= 0:
        _emit(_cfg[0], f"DONE {nodeid} {time.time() - _t0.pop(nodeid, time.time()):.1f}s")


_cfg = [None]


def pytest_sessionstart(session):
    _cfg[0] = session.config
