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
