import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests force kernels through the developer switches (VITA_GEMM_KERNEL, VITA_ATTN64, ...): the library honours them only
# when VITA_DEBUG is set, and reads that flag once per process
os.environ.setdefault("VITA_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch

    return torch.load(os.path.join(GOLDEN, name), weights_only=True)    # plain tensors / numbers / strings only: no code runs on load


PARITY_OUT = os.environ.get("VITA_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "r06_parity.json"))


def record_parity(name, **metrics):
    """Every error a GPU test measures lands in gpurun_out/r06_parity.json (copied to profiles/ after the run): the asserts next
    to the calls sit at <= 1.5 x these values."""
    import json
    try:
        os.makedirs(os.path.dirname(PARITY_OUT), exist_ok=True)
        data = json.load(open(PARITY_OUT)) if os.path.exists(PARITY_OUT) else {}
        data[name] = metrics
        json.dump(data, open(PARITY_OUT, "w"), indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def tol(label, value, limit):
    """assert value < limit, with the measured value recorded next to the limit (gpurun_out/r06_parity.json)."""
    value = float(value)
    node = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].replace("tests/", "")
    key = f"{node}:{label}"
    worst = value
    try:            # the same check inside a loop: keep the largest value seen
        import json
        prev = json.load(open(PARITY_OUT)).get(key) if os.path.exists(PARITY_OUT) else None
        if prev and prev.get("limit") == float(limit):
            worst = max(value, prev.get("value", value))
    except (OSError, ValueError):
        pass
    record_parity(key, value=worst, limit=float(limit))
    assert value < limit, (label, value, limit)
