import os
import sys

import pytest
import torch  # noqa: F401  loaded BEFORE libraytracer_amd.so: torch carries its own HIP runtime, and whichever
#               runtime is loaded second in a process finds no GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference built into oracle/_ref (build container only)")


@pytest.fixture(scope="session")
def gpu():
    """One device context for the whole GPU session (the library, not torch)."""
    import raytracer_amd as rta
    ctx = rta.Context(0)
    yield ctx
    ctx.close()
