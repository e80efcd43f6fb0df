import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    # every GPU test runs against both kernel implementations behind the C ABI
    if metafunc.definition.get_closest_marker("gpu") is not None and "rdyhip_kernel" in metafunc.fixturenames:
        metafunc.parametrize("rdyhip_kernel", ["tiled", "cell"], indirect=True)


@pytest.fixture(autouse=True)
def rdyhip_kernel(request):
    """Selects the kernel variant read by rdyhip_create (RDYHIP_KERNEL)."""
    variant = getattr(request, "param", None)
    old = os.environ.get("RDYHIP_KERNEL")
    if variant is not None:
        os.environ["RDYHIP_KERNEL"] = variant
    yield variant
    if variant is not None:
        if old is None:
            os.environ.pop("RDYHIP_KERNEL", None)
        else:
            os.environ["RDYHIP_KERNEL"] = old
