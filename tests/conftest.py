import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))


# Interop note: PyTorch-ROCm bundles its own libamdhip64, libmozjpeg_hip.so links the system ROCm one.
# Both can live in one process only if torch's runtime is initialised FIRST (the later one reuses the
# already-initialised driver state); the other order leaves torch without devices.  bench.py imports
# torch first for the same reason.  Tests that never touch torch are unaffected.
try:  # noqa: SIM105
    import torch as _torch
    _torch.cuda.is_available()
except Exception:  # torch absent or no GPU: nothing to order
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(HERE, "golden", "goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_images():
    from cases import images
    return images()
