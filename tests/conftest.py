import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; build container only)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc_net
    orc_net.lib()
    return orc_net


@pytest.fixture(scope="session")
def dk():
    import darknet_amd
    if not os.path.exists(darknet_amd.LIB_PATH):
        darknet_amd.build()
    darknet_amd.lib()
    return darknet_amd


@pytest.fixture(scope="session")
def gpu(dk):
    if not dk.have_gpu():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box (-m gpu)")
    dk.lib().cuda_set_device(0)
    return dk
