import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Most GPU tests read a workspace's buffers with the batch's exact (S, T') shape; shape bucketing (padding S / T' up to a
# multiple of Engine.shape_bucket, default 2) is covered by tests/test_gpu_shapes.py, which compares bucketed against exact.
os.environ.setdefault("VMMT_SHAPE_BUCKET", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _gpu_quiet_between_tests(request):
    """The engine runs part of a step on streams of its own (side / AUX / COMM).  A test that drops its engine while those still run would
    hand their buffers back to torch's caching allocator, and the next test's tensors could be written by the previous test's kernels:
    every GPU test ends with the device idle."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
