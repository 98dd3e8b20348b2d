import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- measured-error ledger: parity(label, value, bound) asserts value <= bound AND records the measurement; at session end the
# ledger is printed (-s) and written to gpurun_out/parity_measured.json so that bounds can be kept at "measured x 1.5".
_LEDGER = []


def parity(label, value, bound):
    value = float(value)
    _LEDGER.append((label, value, float(bound)))
    assert value <= bound, f"{label}: measured {value:.3e} > bound {bound:.3e}"


@pytest.fixture(scope="session", autouse=True)
def _parity_ledger():
    yield
    if not _LEDGER:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_measured.json"), "w") as f:
            json.dump([{"label": l, "measured": v, "bound": b} for l, v, b in _LEDGER], f, indent=1)
    except OSError:
        pass
    print("\n[parity ledger] " + "; ".join(f"{l} {v:.2e} (<= {b:.1e})" for l, v, b in _LEDGER))
