import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference in oracle/_ref (skipped when absent)")


@pytest.fixture(scope="session")
def orc():
    import _cabi
    return _cabi.load_orc()


@pytest.fixture(scope="session")
def ref():
    import _cabi
    if not _cabi.have_ref():
        pytest.skip("oracle/_ref/libdsv1ref.so not built (no /root/reference here)")
    return _cabi.load_ref()
