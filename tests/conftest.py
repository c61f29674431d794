import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs the compiled reference in oracle/_ref")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/librawspeed_ref.so not built (no /root/reference here)")
    return Ref()
