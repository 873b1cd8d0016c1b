import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build every native artefact once per session (HIP library, host mirror, CPU oracle)."""
    import __graft_entry__ as g
    g.build()
    return True


def pytest_generate_tests(metafunc):
    """`walk`: every GPU parity test that asks for it runs twice -- "default": the library as shipped and as bench.py times it (intersection
    counters off, the reference's default build, Core/Config.h:4: the 4-wide walks, dense path state, re-trace hand-over); "counting": the
    reference's binary walk with RT_ENABLE_INTERSECTION_COUNTERS-style box / triangle test counters, which are then compared too."""
    if "walk" in metafunc.fixturenames:
        metafunc.parametrize("walk", ["default", "counting"])


@pytest.fixture(autouse=True)
def _device_memory_log(request):
    """RT_TEST_MEMLOG=<file>: appends the free device memory before every test (debugging aid for leaks across tests)."""
    path = os.environ.get("RT_TEST_MEMLOG")
    if path:
        import ctypes
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
            hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
            with open(path, "a") as f:
                f.write("%-90s free %.2f GB\n" % (request.node.name, free.value / 2**30))
        except OSError:
            pass
    yield
