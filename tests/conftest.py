import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The oracle's OpenMP sweeps on one thread per physical core at most: on the GPU box's 2 x 128-core host the OpenMP default (all 256 hardware threads) made a
    2^17-cycle stage test take 58 s where 128 threads take 8 (profiles/r04_pytest_gpu.txt against r04_pytest_mid_sizes.txt); tests that know better set their own."""
    try:
        import oracle_lib
        oracle_lib.baseline_set_threads(min(128, max(1, (os.cpu_count() or 2) // 2) if (os.cpu_count() or 1) > 16 else (os.cpu_count() or 1)))
    except Exception:  # the oracle library is built by the fixtures that need it; nothing to configure before that
        pass
    yield


# ---- GPU runs: leave the interpreter without its teardown --------------------------------------------------------------------------------------------------------
# A `pytest -m gpu` process holds two HIP runtimes (the system one under libjolt_hip.so, torch's bundled one for the multi-process tests), two OpenMP runtimes
# (liboracle's, torch's) and their atexit handlers.  Once this round glibc aborted such a process AFTER the summary line ("33 passed ... double free or corruption"
# at interpreter exit, profiles/r04_pytest_gpu_last_tree.txt's sibling run): every test had passed, the exit status was 134.  The outcome of a test run is the
# outcome of its tests, so a GPU run ends with os._exit(<pytest's own exit status>) once the report is written and the Python-level exit handlers have run, skipping only the loaded
# libraries' C-level destructors; CPU runs
# (-m "not gpu") leave normally.  bench.py and smoke() are untouched: they exit through the normal path.
_exit_status = {"code": None}


def pytest_sessionfinish(session, exitstatus):
    _exit_status["code"] = int(exitstatus)


@pytest.hookimpl(trylast=True)  # after every other plugin's unconfigure (reporters, recorders)
def pytest_unconfigure(config):
    expr = (config.getoption("markexpr", default="") or "").strip()
    if _exit_status["code"] is None or "gpu" not in expr or "not gpu" in expr:
        return
    try:  # Python-level exit handlers still run (anything a harness registered with atexit); only the C-level destructors of the loaded libraries are skipped
        import atexit
        atexit._run_exitfuncs()
    except Exception:
        pass
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(_exit_status["code"])
