"""pytest configuration: registers the `gpu` marker and puts the package directory on sys.path.

`-m "not gpu"` tests run in the CPU-only build container; `-m gpu` tests need a real MI355X and call the HIP kernels
through the C ABI (they fail loudly if the extension is missing — there is no CPU fallback)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "llava-plus-codebase_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` was selected but no GPU is visible")
    return torch.device("cuda:0")
