import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session", autouse=True)
def _library_present():
    """A source-only checkout has no lib/libansfm.so yet: compile it once (hipcc cross-compiles without a GPU).  An
    existing library is left alone -- a snapshot copy does not keep modification times."""
    import archnemesis_dist_amd as pkg
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
