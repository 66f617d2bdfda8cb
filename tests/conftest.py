import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        from rag_inference_pipeline_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _has_gpu():
        pytest.fail("this test is marked gpu but no HIP device / librag_amd.so is usable")
