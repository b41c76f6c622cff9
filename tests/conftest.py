"""pytest configuration: registers the `gpu` marker and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors / vs the compiled reference (when oracle/_ref exists),
                 host logic, C-ABI symbol checks.  No GPU needed.
`-m gpu`       : parity tests proper: the HIP path (through the C ABI) vs the oracle.
"""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ref():
    from tests import oracle_lib
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libref_oracle.so not built (needs /root/reference)")
    return lib


@pytest.fixture(scope="session")
def lib():
    import imagemosaicing_amd as im
    return im
