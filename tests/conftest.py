import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm GPU; none visible (no CPU fallback exists)")
    return torch.device("cuda:0")


def record_measurement(name: str, **values) -> None:
    """Achieved errors the GPU tests measure on the way are appended to
    gpurun_out/test_measurements.jsonl (scratch; the ones tolerances are derived from are
    copied into profiles/), so that a tolerance can be checked against what was measured."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_measurements.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, **{k: float(v) for k, v in values.items()}}) + "\n")
    except OSError:
        pass
