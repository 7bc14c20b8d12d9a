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


def pytest_addoption(parser):
    parser.addoption("--simt", action="store_true", default=False,
                     help="development aid for a container without a GPU: run the -m gpu tests against the kernel SOURCES "
                          "executed by the lock-step wave64 emulator (tools/simt) instead of libmozjpeg_hip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if config.getoption("--simt"):
        # test-side switch only: the package itself knows no other library than libmozjpeg_hip.so
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools", "simt"))
        import build_simt
        import mozjpeg_amd
        mozjpeg_amd.LIB_PATH = build_simt.build()
        os.environ.setdefault("SIMT_STRICT", "1")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--simt"):
        skip = pytest.mark.skip(reason="needs a real device (torch tensors / the shim libraries linked to libmozjpeg_hip.so)")
        for it in items:
            if "torch" in it.name or "tensor" in it.name or "device_batch" in it.name or "device_entry" in it.name or "two_rank" in it.name or it.fspath.basename in ("test_gpu_dropin.py", "test_standalone_api.py"):
                it.add_marker(skip)


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(HERE, "golden", "goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_images():
    from cases import images
    return images()
