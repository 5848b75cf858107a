import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _on_gpu_box():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:                                               # noqa: BLE001
        return False


def pytest_report_header(config):
    from tests import reflib
    return "libzstd 1.5.7 checker: %s" % (reflib.REF_KIND if reflib.have_ref() else "NONE (GPU tests that need it fail, CPU tests skip)")


@pytest.fixture(scope="session")
def ref():
    """libzstd 1.5.7 itself: the reference build (oracle/_ref), else the image's copy. With neither, a GPU run FAILS (a clone-based run must
    not quietly lose its parity tests); the CPU suite skips."""
    from tests import reflib
    if not reflib.have_ref():
        if _on_gpu_box():
            pytest.fail("no libzstd 1.5.7 to check against (oracle/_ref not built and no copy in the image)")
        pytest.skip("no libzstd 1.5.7 available (oracle/_ref/libzstd_ref.so needs /root/reference)")
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
