import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build(ref=True)
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref_oracle():
    """The reference's own objects (oracle/_ref); skipped when it was never built."""
    from oracle import pyoracle
    pyoracle.build(ref=True)
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return pyoracle.Oracle(ref=True)
