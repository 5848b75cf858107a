import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def ref():
    from tests import reflib
    if not reflib.have_ref():
        pytest.skip("oracle/_ref/libzstd_ref.so not built (needs /root/reference)")
    return reflib.RefZstd()


@pytest.fixture(scope="session")
def oracle():
    from tests import reflib
    if not reflib.have_oracle():
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return reflib.Oracle()


@pytest.fixture(scope="session")
def corpus():
    from tests.corpus import Corpus
    return Corpus()
